set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp7.log
: > $O
LIB=vaura_amd/csrc/libvaura_hip.so
WT=vaura_amd/csrc/libvaura_hip_wt.so
echo "== h2: batched (0) vs unbatched (16384) two-plane weights" >> $O
timeout 600 tools/pmc_driver $LIB --time 5 --flags 0,16384 --weights h2 >> $O 2>&1
echo "== write-through output stores (experiment build): h1, h2" >> $O
timeout 600 tools/pmc_driver $LIB --time 5 --flags 0 --weights h1 >> $O 2>&1
timeout 600 tools/pmc_driver $WT --time 5 --flags 0 --weights h1 >> $O 2>&1
timeout 600 tools/pmc_driver $WT --time 5 --flags 0 --weights h2 >> $O 2>&1
echo "== mfma driver smoke" >> $O
timeout 300 tools/mfma_driver $LIB codec 8 1 >> $O 2>&1
timeout 300 tools/mfma_driver $LIB avclip 8 >> $O 2>&1
echo "== counters available" >> $O
rocprofv3 -L 2>/dev/null | grep -E "MFMA|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|LDS_BANK|LDS_IDX|GRBM_GUI|SQ_WAVES|SQ_INSTS_VALU " | sed 's/^ *//' | sort -u | head -40 >> $O
cat $O
