cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_generate.py -x -q -m gpu -k "one_launch_mlp or two_row or generate_cases or prompt_continuation or random_depth or unrounded_checkpoint_against" > gpurun_out/t3.log 2>&1
echo "pytest rc $?" >> gpurun_out/t3.log
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread
rm -f gpurun_out/att_ab.log
for w in h2 h1; do
  timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 7 --flags 0:0,0:4 --rows 16 --weights $w 2>&1 | grep "flags" >> gpurun_out/att_ab.log
done
timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 5 --flags 0:0,0:4 --rows 8 --weights h2 2>&1 | grep "flags" >> gpurun_out/att_ab.log
for t in "" q0 q1 q3 q4 p1 p6 p9; do
  for w in h2 h1; do
    echo "lib${t:+_}$t $w: $(timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip${t:+_}$t.so --time 5 --flags 0 --rows 32 --weights $w 2>&1 | grep 'loop of 228' | cut -c1-120)" >> gpurun_out/att_ab.log
  done
done
tail -5 gpurun_out/t3.log; cat gpurun_out/att_ab.log
