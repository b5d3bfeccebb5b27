set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp3.log
: > $O
LIB=vaura_amd/csrc/libvaura_hip.so
echo "== any-order launch probe" >> $O
timeout 120 tools/microbench/anyorder_probe >> $O 2>&1
echo "== A/B 16-wave workgroups: flags 0, 16384 (w2), 32768 (w13), 49152 (both), bf16" >> $O
timeout 600 tools/pmc_driver $LIB --time 5 --flags 0,16384,32768,49152 >> $O 2>&1
echo "== same, f32 storage: 0, 16384" >> $O
timeout 600 tools/pmc_driver $LIB --time 5 --flags 0,16384 --weights f32 >> $O 2>&1
echo "== stamps bf16" >> $O
timeout 300 tools/pmc_driver vaura_amd/csrc/libvaura_hip_stamps.so --stamps gpurun_out/r03/stamps_bf16.bin --steps 6 --pos0 100 >> $O 2>&1
timeout 300 tools/pmc_driver vaura_amd/csrc/libvaura_hip_stamps.so --stamps gpurun_out/r03/stamps_f32.bin --steps 6 --pos0 100 --weights f32 >> $O 2>&1
python tools/stamp_report.py gpurun_out/r03/stamps_bf16.bin gpurun_out/r03/stamps_bf16.json >> $O 2>&1
python tools/stamp_report.py gpurun_out/r03/stamps_f32.bin gpurun_out/r03/stamps_f32.json >> $O 2>&1
timeout 900 python -m pytest tests/test_gpu_plugins.py tests/test_gpu_avclip.py -x -q 2>&1 | tail -5 >> $O
cat $O
