"""What the fp32 row-major output (and residual input) of nd_gemm_split costs against the frag32b3 image store, per ViT shape at M = 6272:
a lane of the MFMA accumulator holds 4 consecutive output columns of ONE row and 16 lanes hold 16 different rows, so a wave's fp32 store
instruction touches 16 rows x 64 B; its image store writes 512 contiguous bytes.  GPU only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nested_diffusion_amd import ops

def timed(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3

M = 6272
g = torch.Generator().manual_seed(0)
for name, K, N in (("qkv", 768, 2304), ("proj", 768, 768), ("fc2", 3072, 768)):
    xs = ops.split_rows(torch.randn(M, K, generator=g).cuda()); ws = ops.split_rows((torch.randn(N, K, generator=g) / K ** 0.5).cuda())
    b = torch.randn(N, generator=g).cuda(); r = torch.randn(M, N, generator=g).cuda()
    rows = [("fp32 out", lambda: ops.gemm_split(xs, ws, b)),
            ("image out only", lambda: ops.gemm_split(xs, ws, b, want_out=False, want_split=True)),
            ("fp32 out + residual", lambda: ops.gemm_split(xs, ws, b, residual=r)),
            ("image out + residual", lambda: ops.gemm_split(xs, ws, b, residual=r, want_out=False, want_split=True))]
    for label, fn in rows:
        print(f"{name:5s} K={K:5d} N={N:5d}  {label:22s} {timed(fn):7.1f} us", flush=True)
