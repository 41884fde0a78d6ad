#!/bin/bash
# Round profile artefacts (run on the GPU box): bench lines (headline + secondary), rocprofv3 kernel-trace stats of bench.py, and the
# HBM-traffic PMC passes (separate runs, kernel-trace only) over the sampler-only bench.  Output under gpurun_out/prof_r03/;
# copy the summaries into profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default: rc=$?"
python3 bench.py --members 1 --no-cpu-baseline > $OUT/bench_config1_K1.json 2>> $OUT/bench_default.err; echo "K=1: rc=$?"
python3 bench.py --mc 20 --steps 3 --warmup 1 > $OUT/bench_mc20.json 2>> $OUT/bench_default.err; echo "mc=20: rc=$?"
python3 bench.py --mc 20 --batch 70 --steps 2 --warmup 1 > $OUT/bench_mc20_B70.json 2>> $OUT/bench_default.err; echo "mc=20 B=70: rc=$?"
python3 bench.py --dtype f16 > $OUT/bench_f16.json 2>> $OUT/bench_default.err; echo "f16: rc=$?"
python3 bench.py --dtype f16 --timesteps 1000 --steps 5 --warmup 1 > $OUT/bench_f16_T1000.json 2>> $OUT/bench_default.err; echo "f16 T=1000: rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1; echo "stats: rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mc20 -- python3 $GRAFT_REPO_ROOT/bench.py --mc 20 --steps 2 --warmup 1 > $OUT/stats_mc20.log 2>&1; echo "stats mc20: rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 20 32 1 > $OUT/pmc_$c.log 2>&1; echo "pmc $c: rc=$?"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc20_$c -- python3 $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 6 32 20 > $OUT/pmc20_$c.log 2>&1; echo "pmc mc20 $c: rc=$?"
done
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench.csv
find $OUT/stats_mc20 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_mc20.csv
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT > $OUT/traffic_summary.txt; cat $OUT/traffic_summary.txt
