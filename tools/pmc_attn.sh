#!/bin/bash
# PMC passes over the attention kernel alone (tools/bench_attn.py); GPU box.  Output: gpurun_out/pmc_attn/summary.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass_$i -- python3 $GRAFT_REPO_ROOT/tools/bench_attn.py > $OUT/pass_$i.log 2>&1 || { echo "pass $i FAILED"; tail -3 $OUT/pass_$i.log; }
done
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, os, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_attention" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(acc):
    print(f"{c:32s} n={len(acc[c]):4d} mean={sum(acc[c]) / len(acc[c]):16.1f}")
PY
cat $OUT/summary.txt
