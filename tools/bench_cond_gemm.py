"""k_cond_gemm (the LDS-tiled large-M ConditionalLinear kernel) through nd_linear at K = N = 4096: time against the number of
128 x 128 tiles, so that the fixed cost per launch, the cost per tile and the effect of a partly filled last round separate.
GPU only.   python tools/bench_cond_gemm.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import ops

# arguments: M (K = N = 4096) or M,K,N triples
shapes = [tuple(int(u) for u in v.split(",")) if "," in v else (int(v), 4096, 4096) for v in sys.argv[1:]] or \
         [(m, 4096, 4096) for m in (256, 512, 1024, 2048, 3072, 4096, 6144, 640, 1400)]
for M, K, N in shapes:
    w = ops.PackedWeight(torch.randn(N, K, device="cuda") / K ** 0.5)
    b = torch.randn(N, device="cuda")
    sc = torch.rand(N, device="cuda") + 0.5
    x = torch.randn(M, K, device="cuda")
    for _ in range(2):
        ops.linear(x, w, b, act="softplus", scale=sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.linear(x, w, b, act="softplus", scale=sc)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    fl = 2.0 * M * K * N
    print(f"M={M:5d} K={K} N={N} tiles={tiles:5d} ({tiles / 256:.2f} per CU): {us:8.1f} us (incl. ~{M * K * 4 / 4e6 + 3:.0f} us packing x)  {fl / us / 1e6:6.1f} TFLOP/s", flush=True)
