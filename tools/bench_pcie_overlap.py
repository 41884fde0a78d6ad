"""Where the PCIe-inclusive batch loop loses time against the resident one (bench.py's `pcie_inclusive`).  GPU only.
   python tools/bench_pcie_overlap.py [--steps 20]
Variants, all K = 5, T = 100, B = 32 at config dims, same process:
  A  resident: predict_batch on the library's own input buffer (bench.py's timed loop)
  B  + a device-to-device copy of the batch into that buffer per step
  C  B + an independent pinned H2D copy of 19.3 MB on a side stream per step (no dependency: pure overlap cost)
  D  the runner's loader loop (runner._rank_batches) with pageable batches: staging memcpy + side-stream H2D.  Since nd_set_input_flag the loop
     copies straight into the library's input buffer once the running batch has read it (no perturbation flags here); the figures kept in
     profiles/r06_pcie_overlap.txt are those of the earlier staged loop (two device buffers, copy released at the graph boundary) = variant H
  E  D with the batches already pinned (no staging memcpy)
  F / G / H  the synchronisation pieces of D one at a time (see the functions)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
args = argparse.Namespace(batch=32, members=5, timesteps=100, mc=1, dtype="f32")
dev = torch.device("cuda", 0)
runner, cfg = bench.build_runner(args, dev)
eng = runner.engine
eng.seed(1234, first_image=0)
B = 32
images = eng.batch_buffers(B, 1, 100, (3, 224, 224))["images"]
from nested_diffusion_amd import synthetic
images.copy_(synthetic.images(B, seed=1234, device=dev))
other = images.clone()
host = images.cpu()
pinned = host.pin_memory()
tgt = torch.zeros(B, dtype=torch.int64)
side = torch.cuda.Stream(dev)
sink = torch.empty_like(images)


def timeit(fn, n):
    fn(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def A(n):
    for _ in range(n):
        runner.predict_batch(images, clone=False)


def Bv(n):
    for _ in range(n):
        runner.predict_batch(other, clone=False)


def C(n):
    for _ in range(n):
        with torch.cuda.stream(side):
            sink.copy_(pinned, non_blocking=True)
        runner.predict_batch(other, clone=False)


def D(n):
    for x, _ in runner._rank_batches([(host, tgt)] * n, 0, B, B):
        runner.predict_batch(x, clone=False)


def E(n):
    for x, _ in runner._rank_batches([(pinned, tgt)] * n, 0, B, B):
        runner.predict_batch(x, clone=False)


def F(n):                     # C + the compute stream waits for the side-stream copy (dependency, no write-after-read wait)
    main = torch.cuda.current_stream(dev)
    for _ in range(n):
        with torch.cuda.stream(side):
            sink.copy_(pinned, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(side)
        main.wait_event(ev)
        runner.predict_batch(other, clone=False)


def G(n):                     # F + the side stream first waits for everything enqueued on the compute stream
    main = torch.cuda.current_stream(dev)
    for _ in range(n):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sink.copy_(pinned, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(side)
        main.wait_event(ev)
        runner.predict_batch(other, clone=False)


def H(n):                     # G, but the copy of the NEXT batch is enqueued before this batch's launch (the loader loop's order)
    main = torch.cuda.current_stream(dev)
    ev = None
    for _ in range(n):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sink.copy_(pinned, non_blocking=True)
            ev_next = torch.cuda.Event(); ev_next.record(side)
        if ev is not None:
            main.wait_event(ev)
        runner.predict_batch(other, clone=False)
        ev = ev_next


def I(n):                     # F + an event recorded on the compute stream per batch (the marker a write-after-read guard needs), nobody waits on it
    main = torch.cuda.current_stream(dev)
    for _ in range(n):
        with torch.cuda.stream(side):
            sink.copy_(pinned, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(side)
        main.wait_event(ev)
        runner.predict_batch(other, clone=False)
        main.record_event()


def J(n):                     # H with THREE buffers: the guard waits for the launch three batches back (long finished), so the copy starts when it is enqueued
    main = torch.cuda.current_stream(dev)
    done, ev = [None, None, None], None
    for i in range(n):
        k = i % 3
        if done[k] is not None:
            side.wait_event(done[k])
        with torch.cuda.stream(side):
            sink.copy_(pinned, non_blocking=True)
            ev_next = torch.cuda.Event(); ev_next.record(side)
        if ev is not None:
            main.wait_event(ev)
        runner.predict_batch(other, clone=False)
        done[(i - 1) % 3] = main.record_event()
        ev = ev_next


for rnd in range(2):
    print(f"round {rnd}: " + "  ".join(f"{name} {timeit(fn, a.steps):.3f} ms" for name, fn in (("A", A), ("B", Bv), ("C", C), ("D", D), ("E", E), ("F", F), ("G", G), ("H", H), ("I", I), ("J", J))))
