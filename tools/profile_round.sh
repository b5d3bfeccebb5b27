#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root.  Usage: tools/profile_round.sh r03
#   1. rocprofv3 kernel stats of the bench command itself (python) -> gpurun_out/prof_<tag>/stats (+ stats_c4 for configs[3])
#   2. the decode-step kernels through the PRODUCT library driven by a plain C++ program (tools/pmc_driver.cpp dlopen()s
#      vaura_amd/csrc/libvaura_hip.so): kernel stats + separate --pmc passes for FETCH_SIZE and WRITE_SIZE, for both storages
#      (h2 = two fp16 planes, what real checkpoints get; h1 = one plane).  rocprofv3 --pmc crashes at start-up under python on this
#      image; the counters need their own passes (MI355X_MICROARCH.md §rocprofv3 PMC slots).  Drivers are rebuilt every time.
#   3. the MFMA-bound stages (codec decode, Segment-AVCLIP extractor, the four GEMMs of a prompt pass per weight storage) through
#      tools/mfma_driver.cpp: kernel stats + MFMA / LDS /
#      HBM counter passes
#   4. in-kernel s_memrealtime stamps of the diagnostic build (wave 0 only, and every wave) -> per-phase json
set -u
TAG=${1:-r03}
PART=${2:-all}          # all | decode (bench stats + the decode-step driver passes) | mfma (codec / extractor / prefill stages + stamps)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
[ "$PART" = "mfma" ] || rm -rf $OUT
mkdir -p $OUT
exec >> $OUT/round_$PART.log 2>&1
# after every profiler call: drop what the summariser does not read (per-dispatch traces, databases) — a killed run must still fit gpurun's 64 MiB
clean() { find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete; }
LIB=$ROOT/vaura_amd/csrc/libvaura_hip.so
DRV=$ROOT/tools/pmc_driver
MDRV=$ROOT/tools/mfma_driver
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 $ROOT/tools/pmc_driver.cpp -o $DRV -ldl -lpthread || exit 1
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 $ROOT/tools/mfma_driver.cpp -o $MDRV -ldl || exit 1
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != "mfma" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-second --no-plugin --no-extra-configs > $OUT/bench_stats.log 2>&1
tail -n 1 $OUT/bench_stats.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -- python3 $ROOT/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-second --no-plugin > $OUT/bench_stats_c4.log 2>&1
tail -n 1 $OUT/bench_stats_c4.log | cut -c1-200
# round 6: the counter passes SAMPLE the whole loop — 24 cache lengths 10, 19, ..., 217 (mean 113.5 = the loop's mean) — instead of positions
# 100..123; all 228 x 76 dispatches under rocprofv3 counters take ~15 min per pass (first attempt: killed at 45 min)
for W in h2 h1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/drv_stats_$W -- $DRV $LIB --weights $W --steps 24 --pos0 10 --stride 9 > $OUT/drv_stats_$W.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/drv_${C}_$W -- $DRV $LIB --weights $W --steps 24 --pos0 10 --stride 9 > $OUT/drv_${C}_$W.log 2>&1
  done
  tail -n 1 $OUT/drv_stats_$W.log
  clean
done
# configs[4]'s per-GPU shape: fp8 weights against the hi activation plane, fp16 K / V, 32 rows
W=fp8h_rows32
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/drv_stats_$W -- $DRV $LIB --weights fp8h --kv f16 --rows 32 --steps 24 --pos0 10 --stride 9 > $OUT/drv_stats_$W.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/drv_${C}_$W -- $DRV $LIB --weights fp8h --kv f16 --rows 32 --steps 24 --pos0 10 --stride 9 > $OUT/drv_${C}_$W.log 2>&1
done
tail -n 1 $OUT/drv_stats_$W.log
# 17..32 decoder rows (the reference's default batch 16 under CFG): the two-row-block instances, two planes
W=h2_rows32
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/drv_stats_$W -- $DRV $LIB --weights h2 --rows 32 --steps 24 --pos0 10 --stride 9 > $OUT/drv_stats_$W.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/drv_${C}_$W -- $DRV $LIB --weights h2 --rows 32 --steps 24 --pos0 10 --stride 9 > $OUT/drv_${C}_$W.log 2>&1
done
tail -n 1 $OUT/drv_stats_$W.log
clean
fi
if [ "$PART" != "decode" ]; then
for M in codec avclip prefill_h2 prefill_h1; do
  case $M in prefill_h2) A="prefill 8 4";; prefill_h1) A="prefill 8 3";; *) A="$M 8";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mfma_stats_$M -- $MDRV $LIB $A > $OUT/mfma_stats_$M.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma_pmcA_$M -- $MDRV $LIB $A > $OUT/mfma_pmcA_$M.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/mfma_pmcB_$M -- $MDRV $LIB $A > $OUT/mfma_pmcB_$M.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/mfma_${C}_$M -- $MDRV $LIB $A > $OUT/mfma_${C}_$M.log 2>&1
  done
  tail -n 1 $OUT/mfma_stats_$M.log
  clean
done
cd $ROOT
S=vaura_amd/csrc/libvaura_hip_stamps.so
if [ -f $S ]; then
  for W in h2 h1; do
    $DRV $S --stamps $OUT/st_w0_$W.bin --steps 6 --pos0 100 --weights $W > $OUT/stamps_$W.log 2>&1
    python3 tools/stamp_report.py $OUT/st_w0_$W.bin $OUT/stamps_wave0_$W.json >> $OUT/stamps_$W.log 2>&1
    PMC_STAMP_ALL_WAVES=1 $DRV $S --stamps $OUT/st_all_$W.bin --steps 3 --pos0 100 --weights $W >> $OUT/stamps_$W.log 2>&1
    python3 tools/stamp_report.py $OUT/st_all_$W.bin $OUT/stamps_allwaves_$W.json >> $OUT/stamps_$W.log 2>&1
    rm -f $OUT/st_*_$W.bin
    tail -n 8 $OUT/stamps_$W.log
  done
fi
fi
cd $ROOT
# summarise HERE (the per-dispatch counter tables of the 228-step passes are tens of MB each): gpurun_out/profiles_<tag>/ is what travels back
python3 tools/summarize_profile.py $TAG $ROOT/gpurun_out/profiles_$TAG > $OUT/summary_$PART.log 2>&1; tail -n 40 $OUT/summary_$PART.log
# gpurun merges at most 64 MiB back: keep the stats tables, drop the per-dispatch traces, counter tables and databases
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT; find $OUT -name "*.csv" | wc -l
