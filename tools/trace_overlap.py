"""Reads a rocprofv3 --kernel-trace CSV of bench.py and reports, for the conditioner part of the LAST batch, which kernels ran
on which queue and how much of their time overlapped with kernels of another queue.
usage: python3 tools/trace_overlap.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
# the last batch: walk back from the end to the last patchify kernel (first kernel of compute_guiding_prediction)
starts = [i for i, e in enumerate(ev) if "k_patchify" in e[2]]
i0 = starts[-1]
batch = ev[i0:]
t0 = batch[0][0]
# conditioner = until the first step-head / sampler kernel
stop = next((k for k, e in enumerate(batch) if "k_step" in e[2] or "k_init" in e[2]), len(batch))
cond = batch[:stop]
print(f"{path}\nlast batch: {len(batch)} kernels, conditioner part {len(cond)} kernels, {(cond[-1][1] - t0) / 1e3:.1f} us")
queues = sorted({e[3] for e in cond})
busy = {}
for q in queues:
    busy[q] = sum(e[1] - e[0] for e in cond if e[3] == q)
    print(f"queue {q}: {sum(1 for e in cond if e[3] == q)} kernels, busy {busy[q] / 1e3:.1f} us")
# overlap between queues: sweep
pts = []
for s, e, n, q, st in cond:
    pts.append((s, 1, q)); pts.append((e, -1, q))
pts.sort()
active = {q: 0 for q in queues}
last = pts[0][0]
both = anyb = 0
for tstamp, d, q in pts:
    n_act = sum(1 for v in active.values() if v > 0)
    if n_act >= 2: both += tstamp - last
    if n_act >= 1: anyb += tstamp - last
    active[q] += d
    last = tstamp
print(f"some kernel running {anyb / 1e3:.1f} us; kernels of two queues running at once {both / 1e3:.1f} us")
for s, e, n, q, st in cond:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{q} s{st}  {n}")
