#!/bin/bash
# Board power and shader clock (rocm-smi) while ONE kernel form runs back to back: is the bf16 x 9 GEMM power-limited?  GPU box.
#   bash tools/power_probe.sh      -> gpurun_out/power_probe.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/power_probe.txt
: > $OUT
echo "== idle" >> $OUT; rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|Max" >> $OUT
for which in f32 b9 b9_cond stream; do
  python3 $GRAFT_REPO_ROOT/tools/power_load.py $which 9 > $OUT.$which 2>/dev/null &
  pid=$!
  sleep 5                                   # steady state (the first seconds are imports and set-up)
  for i in 1 2 3 4; do
    echo "== $which sample $i" >> $OUT
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" >> $OUT
    sleep 0.6
  done
  wait $pid
  cat $OUT.$which >> $OUT; rm -f $OUT.$which
done
cat $OUT
