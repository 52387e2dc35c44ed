cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_sam3_e2e.py tests/test_rccl_world1.py tests/test_trainer.py tests/test_bench_contract.py tests/test_sam3_data.py tests/test_ddp_gloo.py -m gpu -q -p no:cacheprovider > gpurun_out/r03b_tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r03b_tests.log
timeout 600 python tools/adapter_sweep.py > gpurun_out/r03b_adapter_sweep.json 2> gpurun_out/r03b_sweep.err; echo "sweep rc=$?"
timeout 600 python tools/wobble_bisect.py > gpurun_out/r03b_wobble_bisect.txt 2> gpurun_out/r03b_wobble.err; echo "wobble rc=$?"
timeout 900 python bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r03b_bench.err
