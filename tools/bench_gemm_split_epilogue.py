"""What the epilogue options of nd_gemm_split cost at the fc1 shape (M 6272, K 768, N 3072): bias only, + erf-GELU, + frag32b3 store.  GPU only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nested_diffusion_amd import ops

def timed(fn, reps=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3

M, K, N = 6272, 768, 3072
g = torch.Generator().manual_seed(0)
xs = ops.split_rows(torch.randn(M, K, generator=g).cuda()); ws = ops.split_rows((torch.randn(N, K, generator=g) / K ** 0.5).cuda())
b = torch.randn(N, generator=g).cuda()
for act in (None, "gelu"):
    for want_out, want_split in ((True, False), (False, True), (True, True)):
        us = timed(lambda: ops.gemm_split(xs, ws, b, act=act, want_out=want_out, want_split=want_split))
        print(f"act={act!s:5} fp32 out={want_out!s:5} frag32b3 out={want_split!s:5}: {us:7.1f} us  ({2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s of useful fp32 flop)", flush=True)

# the five ViT Linear shapes at B = 32 (M = 6272): half-tile tail (default) against whole remainder tiles (ND_B9_WHOLE_TAIL=1 in the environment)
print("tail form:", "whole remainder tiles" if os.environ.get("ND_B9_WHOLE_TAIL") else "half tiles / k-slabs (product)")
for name, K_, N_ in (("patch-embed", 768, 768), ("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)):
    xs_ = ops.split_rows(torch.randn(M, K_, generator=g).cuda()); ws_ = ops.split_rows((torch.randn(N_, K_, generator=g) / K_ ** 0.5).cuda())
    bb = torch.randn(N_, generator=g).cuda()
    us = timed(lambda: ops.gemm_split(xs_, ws_, bb))
    print(f"  {name:12s} K={K_:5d} N={N_:5d}: {us:7.1f} us  ({2.0 * M * N_ * K_ / us * 1e-6:6.1f} TFLOP/s)", flush=True)
