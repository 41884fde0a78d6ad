"""Summarise the passes of tools/pmc_mfma.sh: per kernel the mean of every counter per dispatch and the derived MFMA-busy
fraction.  usage: python tools/pmc_mfma.py <gpurun_out/pmc_mfma>  -> CSV on stdout

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs): the share of SIMD-cycles, over the
dispatch's wall time, in which a matrix instruction was executing (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts
cycles, summed over the SIMDs; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs).  A kernel that keeps one f32 MFMA
(v_mfma_f32_16x16x4_f32: 32 cycles, back to back) in flight on every SIMD for its whole duration reads 1.0."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
WANT = ("k_attention", "k_gemm_nt", "k_gemm_b9", "k_b9_fixup", "k_skinny", "k_cond_gemm", "k_step_head", "k_gemm_fixup", "k_layernorm")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "")
            if not any(w in k for w in WANT):
                continue
            key = k.split("(")[0].replace("void ", "").strip()
            if key.startswith("_Z9k_gemm_b9"):       # rocprofv3 leaves this one mangled: template arguments FA, FB, WN, WM, NS
                import re
                key = "k_gemm_b9<" + ", ".join(re.findall(r"Li(\d+)E", key)[:5]) + (", att" if "Lb1E" in key else "") + ">"    # att: the qkv Linear writing attention images
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
failed = []
fp = os.path.join(root, "failed_passes.txt")
if os.path.exists(fp):
    failed = open(fp).read().split()
names = sorted({c for k in acc for c in acc[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches", "mfma_busy_frac"] + names)
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    n = max(len(v) for v in acc[k].values())
    frac = ""
    if m.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        frac = "%.4f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4))
    w.writerow([k, n, frac] + ["%.1f" % m[c] if c in m else "" for c in names])
if failed:
    w.writerow(["FAILED PASSES"] + failed)
