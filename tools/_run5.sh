cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_generate.py -x -q -m gpu -s -k "trained_like or handoff_timeout or range_guard" > gpurun_out/t5.log 2>&1
echo "pytest rc $?" >> gpurun_out/t5.log
tail -12 gpurun_out/t5.log
bash tools/profile_round.sh r05
tail -30 gpurun_out/prof_r05/round.log
