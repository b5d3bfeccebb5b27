set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp2.log
: > $O
timeout 600 python tools/time_loop_parts.py >> $O 2>&1
timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_generate.py::test_configs4_per_gpu_shape_fp8_weights_and_mx8_codec tests/test_gpu_plugins.py tests/test_gpu_avclip.py -x -q -s 2>&1 | tail -40 >> $O
cat $O
