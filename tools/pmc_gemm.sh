#!/bin/bash
# PMC passes over the ViT GEMM kernels (k_gemm_b9 = the product path, k_gemm_nt = the f32-input MFMA form), one shape at a time
# (tools/bench_gemm_vit.py with ND_GEMM_ONLY); GPU box.
# Counters in their own runs, kernel-trace only.  Output: gpurun_out/pmc_gemm/summary.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in qkv fc2; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32"; do
    i=$((i+1))
    ND_GEMM_ONLY=$shape rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${shape}_$i -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm_vit.py > $OUT/${shape}_$i.log 2>&1 || { echo "pass $shape $i FAILED"; tail -3 $OUT/${shape}_$i.log; }
  done
done
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, os, sys, collections
for shape in ("qkv", "fc2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(sys.argv[1], shape + "_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for kern in ("k_gemm_b9", "k_gemm_nt"):
                if kern in r["Kernel_Name"]:
                    acc[(kern, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for kern in ("k_gemm_b9", "k_gemm_nt"):
        print(f"== {kern}, shape {shape} (M = 6272) ==")
        a = {c: v for (k, c), v in acc.items() if k == kern}
        for c in sorted(a):
            print(f"{c:32s} n={len(a[c]):4d} mean={sum(a[c]) / len(a[c]):16.1f}")
        if a.get("GRBM_GUI_ACTIVE") and a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            g = sum(a["GRBM_GUI_ACTIVE"]) / len(a["GRBM_GUI_ACTIVE"]); m = sum(a["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(a["SQ_VALU_MFMA_BUSY_CYCLES"])
            print(f"mfma_busy_frac = {m / (g / 8.0 * 256 * 4):.4f}")
PY
cat $OUT/summary.txt
for d in $OUT/qkv_* $OUT/fc2_*; do [ -d $d ] && rm -rf $d; done     # raw counter CSVs: ~30 MB per pass
