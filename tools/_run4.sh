cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread
rm -f gpurun_out/thr_ab.log
for rep in 1 2 3; do
for t in "" q4 p6 q4p6 q3; do
  for w in h2 h1; do
    echo "rep $rep lib${t:+_}$t $w: $(timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip${t:+_}$t.so --time 5 --flags 0 --rows 32 --weights $w 2>&1 | grep 'loop of 228' | cut -c30-110)" >> gpurun_out/thr_ab.log
  done
done
done
cat gpurun_out/thr_ab.log
