set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp5.log
: > $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -s 2>&1 | tail -8 >> $O
S=vaura_amd/csrc/libvaura_hip_stamps.so
for W in bf16 f32; do
  timeout 300 tools/pmc_driver $S --stamps gpurun_out/r03/st_w0_$W.bin --steps 6 --pos0 100 --weights $W >> $O 2>&1
  python tools/stamp_report.py gpurun_out/r03/st_w0_$W.bin gpurun_out/r03/stamps_wave0_$W.json >> $O 2>&1
  PMC_STAMP_ALL_WAVES=1 timeout 300 tools/pmc_driver $S --stamps gpurun_out/r03/st_all_$W.bin --steps 3 --pos0 100 --weights $W >> $O 2>&1
  python tools/stamp_report.py gpurun_out/r03/st_all_$W.bin gpurun_out/r03/stamps_allwaves_$W.json >> $O 2>&1
done
rm -f gpurun_out/r03/st_*.bin
timeout 1500 python bench.py > gpurun_out/r03/bench_a.json 2> gpurun_out/r03/bench_a.err
tail -c 3000 gpurun_out/r03/bench_a.json >> $O
cat $O
