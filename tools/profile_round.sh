#!/bin/bash
# Round profile artefacts (run on the GPU box): kernel-trace stats of bench.py, then HBM-traffic PMC passes (separate runs,
# kernel-trace only) over the sampler-only bench.  Output under gpurun_out/prof_*; copy the summaries into profiles/.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/prof_pmc_$c -- python3 $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 20 32 1 > $OUT/prof_pmc_$c.log 2>&1
  echo "pmc $c done"
done
find $OUT/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
tail -1 $OUT/prof_stats.log > $OUT/bench_profiled.json
