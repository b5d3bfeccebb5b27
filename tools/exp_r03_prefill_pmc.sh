#!/bin/bash
# MFMA / LDS / HBM counters of the prefill GEMMs only (the prefill part of tools/profile_round.sh)
cd $GRAFT_REPO_ROOT; ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof_r03; mkdir -p $OUT
LIB=$ROOT/vaura_amd/csrc/libvaura_hip.so; MDRV=$ROOT/tools/mfma_driver
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 $ROOT/tools/mfma_driver.cpp -o $MDRV -ldl || exit 1
cd /tmp && export TMPDIR=/tmp
for M in prefill_h2 prefill_h1; do
  case $M in prefill_h2) A="prefill 8 4";; prefill_h1) A="prefill 8 3";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mfma_stats_$M -- $MDRV $LIB $A > $OUT/mfma_stats_$M.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma_pmcA_$M -- $MDRV $LIB $A > $OUT/mfma_pmcA_$M.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/mfma_pmcB_$M -- $MDRV $LIB $A > $OUT/mfma_pmcB_$M.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/mfma_${C}_$M -- $MDRV $LIB $A > $OUT/mfma_${C}_$M.log 2>&1
  done
  tail -n 1 $OUT/mfma_stats_$M.log
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
