"""Compile the k_skinny translation units to ISA and flag k_skinny main loops whose waits degraded to vmcnt(0) only (a pending flat_load
or an uncountable load before the loop does that; the software pipeline then collapses).  CPU only.
   python tools/check_waits.py"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the k_skinny family is instantiated in csrc/nd_skinny_m{0,1,2}.hip (one translation unit per MODE): compiled side by side
from concurrent.futures import ThreadPoolExecutor
def isa(m):
    src = os.path.join(root, "nested_diffusion_amd", "csrc", f"nd_skinny_m{m}.hip")
    out = os.path.join(tempfile.gettempdir(), f"nd_skinny_m{m}_check.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", src, "-S", "--cuda-device-only", "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return open(out).read()
with ThreadPoolExecutor(max_workers=3) as ex:
    txt = "\n".join(ex.map(isa, (0, 1, 2)))
bad = tot = 0
for f in re.split(r"\n(?=_Z[\w]+:\s*;)", txt):
    name = f.split(":")[0]
    m = re.search(r"k_skinnyILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d)", name)
    if not m:
        continue
    L = f.split("\n")
    for i, l in enumerate(L):
        if "Loop Header" not in l:
            continue
        lab = l.split(":")[0]
        body = next((L[i:j + 1] for j in range(i + 1, len(L)) if "s_cbranch" in L[j] and lab in L[j]), None)
        if not body:
            continue
        mf = sum("v_mfma" in x for x in body)
        ld = sum("global_load" in x or "flat_load" in x for x in body)
        if mf < 8 or ld < 6:
            continue
        tot += 1
        waits = [x for x in body if "s_waitcnt vmcnt" in x]
        if all("vmcnt(0)" in w for w in waits) or any("flat_load" in x for x in body):
            bad += 1
            print("DEGRADED:", m.groups(), lab, "mfma", mf, "loads", ld)
print(f"{tot} pipelined loops checked, {bad} degraded")
sys.exit(1 if bad else 0)
