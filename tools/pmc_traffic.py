"""HBM-side traffic per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh.
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB, and on gfx950 FETCH_SIZE counts the 128-byte requests
of a wide coalesced stream at 64 bytes each (MI355X_MICROARCH.md, HBM section) -- hence the factor 2 on reads."""
import collections, csv, glob, os, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(root, "pmc*_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(w in k for w in ("k_skinny", "k_cond_gemm", "k_step_head")):
                continue
            acc[k.split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    n = {c: len(v) for c, v in acc[k].items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        print(f"{k}: FETCH_SIZE {m['FETCH_SIZE']:.1f} KiB (n={n['FETCH_SIZE']}), WRITE_SIZE {m['WRITE_SIZE']:.1f} KiB (n={n['WRITE_SIZE']}) "
              f"-> {(2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024 / 1e6:.1f} MB per launch")
