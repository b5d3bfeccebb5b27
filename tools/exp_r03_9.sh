set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp9.log
: > $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generate.py -q -x -k "codec or dac or avclip" 2>&1 | tail -5 >> $O
timeout 900 python -m pytest tests/test_gpu_avclip.py tests/test_gpu_e2e.py -q -x 2>&1 | tail -3 >> $O
timeout 600 python tools/time_codec.py >> $O 2>&1
timeout 600 python tools/time_avclip.py >> $O 2>&1
timeout 900 python tools/time_sliding_window.py >> $O 2>&1
timeout 900 python tools/bench_longform.py >> $O 2>&1
cat $O
