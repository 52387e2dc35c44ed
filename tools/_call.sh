cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_sam3_e2e.py tests/test_trainer.py tests/test_fp8.py tests/test_vit_parity.py -m gpu -q -p no:cacheprovider > gpurun_out/r03c_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r03c_tests.log
timeout 600 python tools/adapter_sweep.py > gpurun_out/r03c_adapter_sweep.json 2> gpurun_out/r03c_sweep.err; echo "sweep rc=$?"
