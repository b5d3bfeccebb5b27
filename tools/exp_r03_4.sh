set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp4.log
: > $O
LIB=vaura_amd/csrc/libvaura_hip.so
echo "== A/B after the rinv-before-barrier change: flags 0, 32768 (w13 16-wave), bf16" >> $O
timeout 600 tools/pmc_driver $LIB --time 5 --flags 0,32768 >> $O 2>&1
echo "== f32 storage" >> $O
timeout 600 tools/pmc_driver $LIB --time 5 --flags 0 --weights f32 >> $O 2>&1
echo "== stamps" >> $O
timeout 300 tools/pmc_driver vaura_amd/csrc/libvaura_hip_stamps.so --stamps gpurun_out/r03/stamps_bf16.bin --steps 4 --pos0 100 >> $O 2>&1
timeout 300 tools/pmc_driver vaura_amd/csrc/libvaura_hip_stamps.so --stamps gpurun_out/r03/stamps_f32.bin --steps 4 --pos0 100 --weights f32 >> $O 2>&1
python tools/stamp_report.py gpurun_out/r03/stamps_bf16.bin gpurun_out/r03/stamps_bf16.json >> $O 2>&1
python tools/stamp_report.py gpurun_out/r03/stamps_f32.bin gpurun_out/r03/stamps_f32.json >> $O 2>&1
rm -f gpurun_out/r03/stamps_*.bin
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 >> $O
cat $O
