#!/bin/bash
# Board power and shader clock while the sampler graph (head + weight-streaming step blocks) runs; GPU box.
# Samples rocm-smi every 0.5 s for the life of the process and keeps the samples above 400 W (set-up idles at ~260 W).
OUT=$GRAFT_REPO_ROOT/gpurun_out/power_probe_sampler.txt
: > $OUT
for DT in f32 f16; do
  ND_DTYPE=$DT ND_BENCH_REPS=80 python3 $GRAFT_REPO_ROOT/tools/bench_sampler.py 5 1000 32 1 > $OUT.s 2>/dev/null &
  pid=$!
  echo "== sampler, $DT operands (samples above 400 W)" >> $OUT
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' ' | awk '{ for (i = 1; i <= NF; i++) if ($i == "(W):") { if ($(i+1) + 0 > 400) print $0 } }' >> $OUT
    sleep 0.5
  done
  wait $pid; cat $OUT.s >> $OUT; rm -f $OUT.s
done
cat $OUT
