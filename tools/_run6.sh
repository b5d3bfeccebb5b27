cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_generate.py tests/test_gpu_ops.py -x -q -m gpu -k "fp8 or two_row or configs4" > gpurun_out/t6.log 2>&1
echo "pytest rc $?" >> gpurun_out/t6.log
tail -4 gpurun_out/t6.log
timeout 600 python tools/ab_fp8_rowsplit.py 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_fp8.log
timeout 600 python tools/ab_fp8_rowsplit.py 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab_fp8.log
