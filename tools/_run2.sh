cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_generate.py -x -q -m gpu -k "one_launch_mlp or two_row or range_guard" > gpurun_out/t2.log 2>&1
echo "pytest rc $?" >> gpurun_out/t2.log
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread
rm -f gpurun_out/rb2e_ab.log
for w in h2 h1; do
  timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 5 --flags 0:0,0:2,0:3 --rows 32 --weights $w 2>&1 | grep "flags" >> gpurun_out/rb2e_ab.log
  timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 5 --flags 0:0,0:2 --rows 24 --weights $w 2>&1 | grep "flags" >> gpurun_out/rb2e_ab.log
  timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip.so --time 5 --flags 0 --rows 16 --weights $w 2>&1 | grep "flags" >> gpurun_out/rb2e_ab.log
done
tail -5 gpurun_out/t2.log; cat gpurun_out/rb2e_ab.log
