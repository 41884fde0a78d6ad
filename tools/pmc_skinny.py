"""Summarise rocprofv3 --pmc passes for the step GEMM (k_skinny, 256-workgroup launches).
usage: python tools/pmc_skinny.py <dir with pass_*/ ... counter_collection.csv>  -> mean per dispatch of every counter"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "")
            if "k_skinny" not in k:
                continue
            key = k.split("(")[0].replace("void ", "")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:42s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
