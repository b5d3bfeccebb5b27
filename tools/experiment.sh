#!/bin/bash
# ONE entry point for the measurement recipes that used to be tools/exp_r0N_*.sh.  Run ON THE GPU BOX from the repo root:
#     gpurun --timeout 900 -- 'bash tools/experiment.sh <name> [args]'
# Everything lands under gpurun_out/<name>/ (scratch; copy what should be judged into profiles/).
#
#   loop-ab [flags,flags,..] [rounds]   interleaved A/B of kernel variants (vaura_set_debug_flags values; tools/README.md lists the bits) over
#                                       whole 228-step graph-replayed loops of the PRODUCT library, both storages, + per-stage eager averages
#   stamps [flags ..]                   in-kernel s_memrealtime timeline of the one-launch MLP / layer-tail kernels (diagnostic build;
#                                       ALLWAVES=1: every wave records — perturbs heavily)
#   lib-ab [tags ..]                    whole 228-step loops on experiment builds (`python -m vaura_amd.csrc.build --tag T -DX=..` -> libvaura_hip_T.so) next to the
#                                       product library, two rounds, both storages
#   engines [rounds]                    the three measured-negative engines (attention + wo: flag 4096; the layer tail: flag 8; the attention as a fourth
#                                       phase of the one-launch MLP: 0:4), which live in experiment builds only since round 6: builds
#                                       libvaura_hip_engines.so (-DVAURA_EXPERIMENT_ENGINES=1) and times whole loops per flag set on it
#   graph-steps                         decode steps per graph launch (debug flag bits 24..27): 1 / 4 / 12, alternating
#   chains                              the batch as 2 / 4 independent decode chains on separate streams against one chain of all rows
#   plain-stores                        write-through output stores (product) against ordinary ones (build --plain-stores), alternating
#   codec-abl [tags ..]                 codec decode (mfma_driver codec 8, precisions 1 and 4) on experiment builds `python -m vaura_amd.csrc.build --tag T -DVA_CONV_ABL=n`
#                                       (timing ablations of conv_pair_kernel, csrc/dac.hip) next to the product library
#   linear-abl [tags ..]                extractor forward (mfma_driver avclip 8) on experiment builds `python -m vaura_amd.csrc.build --tag T -DVA_LIN_ABL=n`
#                                       (timing ablations of linear_pair_kernel, csrc/dac.hip) next to the product library
#   codec-pmc                           SQ wait / issue / LDS counters of the codec kernels (mfma_driver codec 8 under rocprofv3 --pmc, four passes), per kernel
#   codec-layers                        per-dispatch durations of one codec decode (kernel trace of mfma_driver codec 8)
#   avclip-stats                        per-kernel averages of one extractor forward (mfma_driver avclip 8 under rocprofv3 --stats)
#   prefill-ab                          one later chunk of the sliding-window caller: LDS-DMA prefill GEMM (0) against the register-staged one (131072)
#   prefill-pmc                         MFMA / LDS / HBM counters of the prefill GEMMs only (the prefill part of tools/profile_round.sh)
#   sw-prof [flags ..]                  rocprofv3 kernel stats of one later chunk of the sliding-window caller, per debug-flag set
#   suite                               the whole -m gpu suite, the default bench line, configs[3] / configs[4] / batch-16 bench lines and the
#                                       secondary timings (codec, extractor, sliding window, long form)
set -u
NAME=${1:-}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/${NAME:-none}; mkdir -p $OUT
LIB=$ROOT/vaura_amd/csrc/libvaura_hip.so
HIPCC=/opt/rocm/bin/hipcc
pmc_driver() { [ -x /tmp/pmc_driver ] || $HIPCC -O2 -std=c++17 --offload-arch=gfx950 tools/pmc_driver.cpp -o /tmp/pmc_driver -ldl -lpthread || exit 1; }
mfma_driver() { [ -x /tmp/mfma_driver ] || $HIPCC -O2 -std=c++17 --offload-arch=gfx950 tools/mfma_driver.cpp -o /tmp/mfma_driver -ldl || exit 1; }
bench_line() {   # bench_line <file> <key> ...: selected keys of the last JSON line of a bench record
  python3 - "$@" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in sys.argv[2:]:
    print(k, json.dumps(d.get(k))[:1200])
PY
}

case "$NAME" in
loop-ab)
  pmc_driver
  for w in h2 h1; do
    echo "== weights $w"
    timeout 600 /tmp/pmc_driver $LIB --time ${2:-5} --flags ${1:-0,4} --weights $w 2>&1 | grep "flags\|host enqueue" | tee -a $OUT/loop_ab.log
  done ;;
stamps)
  pmc_driver
  [ -n "${ALLWAVES:-}" ] && export PMC_STAMP_ALL_WAVES=1
  for w in h2 h1; do for f in ${@:-0}; do
    timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip_stamps.so --stamps $OUT/st_${w}_$f.bin --flags $f --weights $w --steps 2 --pos0 100
    python3 tools/engine_stamps.py $OUT/st_${w}_$f.bin | tee $OUT/stamps_${w}_f$f.txt
    rm -f $OUT/st_${w}_$f.bin
  done; done ;;
engines)
  pmc_driver
  python3 -m vaura_amd.csrc.build --tag engines -DVAURA_EXPERIMENT_ENGINES=1 || exit 1
  for w in h2 h1; do
    echo "== weights $w (libvaura_hip_engines.so)"
    timeout 600 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip_engines.so --time ${1:-3} --flags 0,4096,8,0:4 --weights $w 2>&1 | grep "flags\|host enqueue" | tee -a $OUT/engines.log
  done ;;
lib-ab)
  pmc_driver
  for w in h2 h1; do for rep in 1 2; do for t in "" "$@"; do
    echo "$w lib${t:+_}$t: $(timeout 300 /tmp/pmc_driver vaura_amd/csrc/libvaura_hip${t:+_}$t.so --time 5 --flags 0 --weights $w 2>&1 | grep 'loop of 228' | cut -c1-120)"
  done; done; done | tee $OUT/lib_ab.log ;;
graph-steps)
  pmc_driver
  for w in h2 h1; do for k in 1 4 12 1 4 12; do
    echo "$w steps/graph $k: $(timeout 300 /tmp/pmc_driver $LIB --time 5 --flags $((k << 24)) --weights $w 2>&1 | grep 'loop of 228' | cut -c1-110)"
  done; done | tee $OUT/graph_steps.log ;;
chains)
  pmc_driver
  { echo "== 1 chain x 16 rows"; timeout 300 /tmp/pmc_driver $LIB --time 5
    echo "== 2 chains x 8 rows, thread per chain"; timeout 300 /tmp/pmc_driver $LIB --chains 2 --time 5
    echo "== 2 chains x 8 rows, one host thread"; PMC_ONE_THREAD=1 timeout 300 /tmp/pmc_driver $LIB --chains 2 --time 5
    echo "== 4 chains x 4 rows"; timeout 300 /tmp/pmc_driver $LIB --chains 4 --time 5
    echo "== 1 chain x 8 rows alone"; timeout 300 /tmp/pmc_driver $LIB --rows 8 --time 5; } 2>&1 | tee $OUT/chains.log ;;
plain-stores)
  pmc_driver
  for w in h2 h1; do for rep in 1 2; do for lib in libvaura_hip.so libvaura_hip_plain.so; do
    timeout 300 /tmp/pmc_driver vaura_amd/csrc/$lib --time 5 --weights $w > $OUT/t.log 2>&1; echo "$w $lib: $(grep 'loop of 228' $OUT/t.log | cut -c1-110)"
  done; done; done | tee $OUT/plain_stores.log ;;
codec-abl)
  mfma_driver
  for rep in 1 2; do for t in "" ${@:-_cabl1 _cabl2 _cabl4 _cabl3}; do for pr in 1 4; do
    echo "lib$t precision $pr: $(timeout 120 /tmp/mfma_driver vaura_amd/csrc/libvaura_hip$t.so codec 8 $pr 5 2>&1 | grep -o 'last decode [0-9.]* ms')" | tee -a $OUT/codec_abl.log
  done; done; done ;;
linear-abl)
  mfma_driver
  for rep in 1 2; do for t in "" ${@:-_labl1 _labl2 _labl4 _labl8}; do
    echo "lib$t: $(timeout 120 /tmp/mfma_driver vaura_amd/csrc/libvaura_hip$t.so avclip 8 1 3 2>&1 | grep -o 'last forward [0-9.]* ms')" | tee -a $OUT/linear_abl.log
  done; done ;;
codec-pmc)
  mfma_driver; cd /tmp && export TMPDIR=/tmp
  i=0
  for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc$i -- /tmp/mfma_driver $LIB codec 8 ${1:-1} > $OUT/pmc$i.log 2>&1
  done
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for i in range(1, 6):
    for f in glob.glob("$OUT/pmc%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_INST_LEVEL_LDS", "GRBM_GUI_ACTIVE"): n[(k, r["Counter_Name"])] += 1
with open("$OUT/codec_pmc.txt", "w") as out:
    for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
        out.write(k + "\n")
        for name, v in sorted(c.items()):
            out.write("   %-28s %.4g\n" % (name, v))
print(open("$OUT/codec_pmc.txt").read())
PY
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete ;;
codec-layers)
  mfma_driver; cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o c -- /tmp/mfma_driver $LIB codec 8 > $OUT/run.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
with open("$OUT/dispatches.txt", "w") as out:
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out.write(f'{r["Kernel_Name"][:60]:60s} grid {r["Grid_Size_X"]:>8s} {r["Grid_Size_Y"]:>5s} {r["Grid_Size_Z"]:>4s} wg {r["Workgroup_Size_X"]:>4s} {d:9.1f} us\n')
PY
  rm -rf $OUT/trace; tail -45 $OUT/dispatches.txt ;;
avclip-stats)
  mfma_driver; cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- /tmp/mfma_driver $LIB avclip 8 > $OUT/run.log 2>&1
  tail -1 $OUT/run.log
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f'{r["Name"][:60]:60s} {r["Calls"]:>5s} {float(r["AverageNs"])/1e3:9.1f} us {r["Percentage"]:>6s} %')
PY
  rm -rf $OUT/t ;;
prefill-ab)
  for w in h1 h2; do for f in 0 131072 0 131072; do
    VAURA_DEBUG_FLAGS=$f VAURA_WEIGHTS=$w VAURA_PREFILL_PASSES=192 timeout 600 python3 tools/time_sliding_window.py > $OUT/sw_${w}_$f.log 2>&1
    echo "$w flags $f: $(grep 'ms per chunk' $OUT/sw_${w}_$f.log)"
  done; done ;;
prefill-pmc)
  mfma_driver; cd /tmp && export TMPDIR=/tmp
  for M in prefill_h2 prefill_h1; do
    case $M in prefill_h2) A="prefill 8 4";; prefill_h1) A="prefill 8 3";; esac
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mfma_stats_$M -- /tmp/mfma_driver $LIB $A > $OUT/mfma_stats_$M.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma_pmcA_$M -- /tmp/mfma_driver $LIB $A > $OUT/mfma_pmcA_$M.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/mfma_pmcB_$M -- /tmp/mfma_driver $LIB $A > $OUT/mfma_pmcB_$M.log 2>&1
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/mfma_${C}_$M -- /tmp/mfma_driver $LIB $A > $OUT/mfma_${C}_$M.log 2>&1
    done
    tail -n 1 $OUT/mfma_stats_$M.log
  done
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete ;;
sw-prof)
  export VAURA_PREFILL_PASSES=192
  cd /tmp && export TMPDIR=/tmp
  for f in ${@:-0 131072}; do
    export VAURA_DEBUG_FLAGS=$f
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$f -o sw -- python3 $ROOT/tools/time_sliding_window.py > $OUT/sw_$f.log 2>&1
    grep "ms per chunk" $OUT/sw_$f.log
    cp $(find $OUT/trace_$f -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$f.csv; grep "gemm\|prefill\|rope_append" $OUT/kernel_stats_$f.csv | cut -c1-60,100-220
    rm -rf $OUT/trace_$f
  done ;;
suite)
  timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/gpu_tests.log
  timeout 1700 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.err
  bench_line $OUT/bench_default.json value ms_per_step value_storage value_h1_lossless_checkpoint split_ms decode_loop_roofline \
      decode_loop_roofline_h1_lossless_checkpoint roofline kernel_us kernel_frac_of_hbm_peak codec_roofline plugin_surface end_to_end_with_extractor \
      cpu_baseline gpu_over_cpu | tee $OUT/bench_default.txt
  timeout 900 python3 bench.py --workload c4 --no-cpu-baseline --no-plugin > $OUT/bench_c4.json 2>/dev/null
  bench_line $OUT/bench_c4.json value ms_per_step value_h1_lossless_checkpoint decode_loop_roofline | tee $OUT/bench_c4.txt
  timeout 900 python3 bench.py --weights fp8 --codec mx8 --batch 16 --no-cpu-baseline --no-plugin > $OUT/bench_c5.json 2>/dev/null
  bench_line $OUT/bench_c5.json value ms_per_step split_ms | tee $OUT/bench_c5.txt
  timeout 900 python3 bench.py --batch 16 --no-cpu-baseline --no-plugin --no-second > $OUT/bench_b16.json 2>/dev/null
  bench_line $OUT/bench_b16.json value ms_per_step split_ms | tee $OUT/bench_b16.txt
  { timeout 600 python3 tools/time_codec.py; timeout 600 python3 tools/time_avclip.py; timeout 900 python3 tools/time_sliding_window.py
    timeout 900 python3 tools/bench_longform.py; } 2>&1 | tee $OUT/secondary.log ;;
*)
  sed -n 2,22p $0; exit 1 ;;
esac
