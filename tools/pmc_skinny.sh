#!/bin/bash
# PMC passes over the sampler-only bench (one counter group per pass, kernel-trace only).  Run on the GPU box:
#   bash tools/pmc_skinny.sh [f32|f16]
# (the TA/TD counters go two / one / one per pass: all four in one pass exceed what gfx950 can collect -- rocprofv3 error 38)
# A failed pass is recorded in failed_passes.txt and makes the script exit 1 after the remaining passes.
DT=${1:-f32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$DT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ND_DTYPE=$DT
i=0
fail=0
rm -f $OUT/failed_passes.txt
for grp in "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES" \
           "TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" \
           "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUSY" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" \
           "TA_DATA_STALLED_BY_TC_CYCLES" \
           "TD_TD_BUSY" \
           "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_INFLIGHT_MAX" \
           "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass_$i -- python3 $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 12 32 1 > $OUT/pass_$i.log 2>&1 || { echo "pass $i FAILED: $grp"; tail -5 $OUT/pass_$i.log; echo "pass_$i: $grp" >> $OUT/failed_passes.txt; fail=1; }
  echo "pass $i done"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_skinny.py $OUT > $OUT/summary.txt
[ -f $OUT/failed_passes.txt ] && { echo "FAILED PASSES:"; cat $OUT/failed_passes.txt; } >> $OUT/summary.txt
cat $OUT/summary.txt
exit $fail
