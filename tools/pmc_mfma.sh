#!/bin/bash
# MFMA-utilisation counters (rocprofv3 PMC, one counter group per pass, kernel-trace only -- never combined with --stats or
# other trace domains) for the matrix-core kernels of the hot path:
#   conditioner  (tools/bench_cond.py):          k_attention*, k_gemm_b9 (k_gemm_nt with ND_GEMM_F32=mfma_f32)
#   sampler mc=1 (tools/bench_sampler.py 5 6 32 1):   k_skinny
#   sampler mc=20 (tools/bench_sampler.py 5 4 32 20): k_cond_gemm_b9 (k_cond_gemm with ND_STEP_F32_MFMA=1)
# Run on the GPU box:  bash tools/pmc_mfma.sh      -> gpurun_out/pmc_mfma/{summary.csv, *.log}
# A pass that fails makes the script fail (exit 1) after the remaining passes have run; the failure is named in summary.csv.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
fail=0
run_pass() {   # name, counters, program args...
  local name=$1 grp=$2; shift 2
  if rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1; then
    echo "pass $name ok"
  else
    echo "pass $name FAILED (see $name.log)"; tail -3 $OUT/$name.log; echo "$name" >> $OUT/failed_passes.txt; fail=1
  fi
}
rm -f $OUT/failed_passes.txt
# two SQ groups (8 SQ slots per pass on gfx950; GRBM has its own 2)
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
G2="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
run_pass cond_g1 "$G1" $GRAFT_REPO_ROOT/tools/bench_cond.py
run_pass cond_g2 "$G2" $GRAFT_REPO_ROOT/tools/bench_cond.py
run_pass samp1_g1 "$G1" $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 6 32 1
run_pass samp1_g2 "$G2" $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 6 32 1
run_pass samp20_g1 "$G1" $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 4 32 20
run_pass samp20_g2 "$G2" $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 4 32 20
python3 $GRAFT_REPO_ROOT/tools/pmc_mfma.py $OUT > $OUT/summary.csv
cat $OUT/summary.csv
# the per-dispatch counter CSVs are ~30 MB per pass and gpurun copies back at most 64 MiB: keep the summary and the logs only
for d in cond_g1 cond_g2 samp1_g1 samp1_g2 samp20_g1 samp20_g2; do rm -rf $OUT/$d; done
exit $fail
