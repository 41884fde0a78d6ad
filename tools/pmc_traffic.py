"""HBM-side traffic per launch of the step kernels from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, per SHAPE.

    python tools/pmc_traffic.py <dir with pmc_<tag>_<COUNTER>/ subdirectories> [--json profiles/traffic.json --source <name>]

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB, and on gfx950 FETCH_SIZE counts the 128-byte requests of
a wide coalesced stream at 64 bytes each (MI355X_MICROARCH.md, HBM section) -- hence the factor 2 on reads.  Each counter is collected
in its own rocprofv3 run (kernel-trace only).  <tag> names the shape the pass ran at, e.g. f32_M32_K5 (dtype, rows per member =
B * mc, members): bench.py looks its own shape up under that key."""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
source = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else root
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for d in sorted(glob.glob(os.path.join(root, "pmc_*_*"))):
    m = re.match(r"pmc_(.+)_(FETCH_SIZE|WRITE_SIZE)$", os.path.basename(d))
    if not m or not os.path.isdir(d):
        continue
    tag = m.group(1)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(w in k for w in ("k_skinny", "k_cond_gemm", "k_step_head")):
                continue
            acc[tag][k.split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
entries = {}
for tag in sorted(acc):
    print(f"== {tag}")
    per = {}
    for k in sorted(acc[tag]):
        m = {c: sum(v) / len(v) for c, v in acc[tag][k].items()}
        n = {c: len(v) for c, v in acc[tag][k].items()}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            b = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
            per[k] = {"bytes_per_launch": b, "launches_fetch_pass": n["FETCH_SIZE"], "launches_write_pass": n["WRITE_SIZE"]}
            print(f"{k}: FETCH_SIZE {m['FETCH_SIZE']:.1f} KiB (n={n['FETCH_SIZE']}), WRITE_SIZE {m['WRITE_SIZE']:.1f} KiB (n={n['WRITE_SIZE']}) "
                  f"-> {b / 1e6:.1f} MB per launch")
    # the dominant step launches of the shape: the two ConditionalLinear blocks (MODE 0 / 1), profiled launches only (> 10: the
    # set-up launches of other shapes in the same process are left out); the k-split fixup of the tiled kernel rides with its block
    blocks = [v["bytes_per_launch"] for k, v in per.items() if ("k_skinny" in k or re.match(r"k_cond_gemm(_b9)?<", k)) and v["launches_fetch_pass"] > 10]
    fix = [v["bytes_per_launch"] for k, v in per.items() if "k_cond_gemm_fixup" in k or "k_cond_gemm_b9_fixup" in k]
    if blocks:
        entries[tag] = {"step_block_bytes_per_launch": sum(blocks) / len(blocks) + (sum(fix) / len(fix) if fix else 0.0), "kernels": per}
if jpath:
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only) over tools/bench_sampler.py at "
                         "each shape; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, the x2 being the gfx950 correction of MI355X_MICROARCH.md "
                         "(HBM section); FETCH_SIZE counts Infinity-Cache hits too; step_block_bytes_per_launch = mean over the lin2 and "
                         "lin3(+lin4) launches (+ the k-split fixup of k_cond_gemm)",
               "source": source, "key": "<dtype>_M<rows per member = B*mc>_K<members>", "entries": entries}, open(jpath, "w"), indent=1)
    print("wrote", jpath, "with", sorted(entries))
