cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03a_tests.log 2>&1; echo "tests rc=$?" 
tail -30 gpurun_out/r03a_tests.log
timeout 600 python tools/bf16_parity_probe.py > gpurun_out/r03a_bf16_parity_probe.json 2> gpurun_out/r03a_probe.err; echo "probe rc=$?"
timeout 600 python tools/adapter_sweep.py > gpurun_out/r03a_adapter_sweep.json 2> gpurun_out/r03a_sweep.err; echo "sweep rc=$?"
timeout 600 python tools/wobble_bisect.py > gpurun_out/r03a_wobble_bisect.txt 2> gpurun_out/r03a_wobble.err; echo "wobble rc=$?"
tail -5 gpurun_out/r03a_probe.err gpurun_out/r03a_sweep.err gpurun_out/r03a_wobble.err
