cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_generate.py -x -q -m gpu -s -k "two_row_blocks or shipped_defaults or later_chunk or headline or configs3_long or unrounded_checkpoint_matches or full_size_greedy or two_row or one_launch_mlp" > gpurun_out/t1.log 2>&1
echo "pytest rc $?" >> gpurun_out/t1.log
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread
for w in h2 h1; do
  timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 5 --flags 0:0,0:1 --rows 32 --weights $w 2>&1 | grep "flags\|host" >> gpurun_out/rb2_ab.log
  timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 5 --flags 0:0,0:1 --rows 24 --weights $w 2>&1 | grep "flags" >> gpurun_out/rb2_ab.log
done
timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 5 --flags 0 --rows 16 --weights h2 2>&1 | grep "flags" >> gpurun_out/rb2_ab.log
tail -5 gpurun_out/t1.log; cat gpurun_out/rb2_ab.log
