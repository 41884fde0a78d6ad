#!/bin/bash
# Round profile artefacts (run on the GPU box):  R=r05 bash tools/profile_round.sh [bench|stats|pmc ...]   (default: all three)
#   bench  bench lines (headline + secondary shapes)
#   stats  rocprofv3 kernel-trace stats of bench.py (headline, K = 1, f16, mc = 20)
#   pmc    HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel-trace only) over the sampler-only bench at every
#          shape bench.py reports a roofline for -> traffic.json keyed by shape
# Output under gpurun_out/prof_$R/; copy the summaries into profiles/ (tools/pmc_traffic.py writes traffic.json there directly).
R=${R:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
WHAT=${@:-bench stats pmc}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [[ " $WHAT " == *" bench "* ]]; then
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default: rc=$?"
python3 bench.py --members 1 --no-cpu-baseline > $OUT/bench_config1_K1.json 2>> $OUT/bench_default.err; echo "K=1: rc=$?"
python3 bench.py --mc 20 --steps 3 --warmup 1 > $OUT/bench_mc20_B32.json 2>> $OUT/bench_default.err; echo "mc=20: rc=$?"
python3 bench.py --mc 20 --batch 70 --steps 2 --warmup 1 > $OUT/bench_mc20_B70.json 2>> $OUT/bench_default.err; echo "mc=20 B=70: rc=$?"
python3 bench.py --dtype f16 > $OUT/bench_f16_secondary.json 2>> $OUT/bench_default.err; echo "f16: rc=$?"
python3 bench.py --dtype f16 --timesteps 1000 --steps 5 --warmup 1 > $OUT/bench_config4_f16_T1000.json 2>> $OUT/bench_default.err; echo "f16 T=1000: rc=$?"
fi
cd /tmp && export TMPDIR=/tmp
if [[ " $WHAT " == *" stats "* ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1; echo "stats: rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_K1 -- python3 $GRAFT_REPO_ROOT/bench.py --members 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats_K1.log 2>&1; echo "stats K1: rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f16 -- python3 $GRAFT_REPO_ROOT/bench.py --dtype f16 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats_f16.log 2>&1; echo "stats f16: rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mc20 -- python3 $GRAFT_REPO_ROOT/bench.py --mc 20 --steps 2 --warmup 1 > $OUT/stats_mc20.log 2>&1; echo "stats mc20: rc=$?"
for t in "" _K1 _f16 _mc20; do find $OUT/stats$t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench$t.csv; done
fi
if [[ " $WHAT " == *" pmc "* ]]; then
# tag = <dtype>_M<rows per member>_K<members>; bench_sampler.py arguments: K T B mc
run_pmc() { # tag dtype K T B mc
  export ND_DTYPE=$2
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$1_$c -- python3 $GRAFT_REPO_ROOT/tools/bench_sampler.py $3 $4 $5 $6 > $OUT/pmc_$1_$c.log 2>&1; echo "pmc $1 $c: rc=$?"
  done
  unset ND_DTYPE
}
run_pmc f32_M32_K5 f32 5 20 32 1
run_pmc f32_M32_K1 f32 1 20 32 1
run_pmc f16_M32_K5 f16 5 20 32 1
run_pmc f32_M640_K5 f32 5 6 32 20
run_pmc f32_M1400_K5 f32 5 4 70 20
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT --json $GRAFT_REPO_ROOT/gpurun_out/prof_$R/traffic.json --source profiles/${R}_pmc_hbm_traffic.txt > $OUT/${R}_pmc_hbm_traffic.txt; cat $OUT/${R}_pmc_hbm_traffic.txt
fi
