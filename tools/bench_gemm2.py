"""GEMM shape sweeps: python tools/bench_gemm2.py  (env ND_GEMM_TILE etc. apply)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import ops

def bench(M, K, N, reps=20):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    for _ in range(3): ops.gemm_bias_act(x, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ops.gemm_bias_act(x, w, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    t = ((M + 127) // 128) * ((N + 63) // 64)
    print(f"M={M:6d} K={K:5d} N={N:5d} tiles64={t:5d} ({t/256:5.2f}/CU): {us:8.1f} us  {2*M*K*N/us/1e6:7.1f} TFLOP/s", flush=True)

for M in (6272, 6144, 8192):
    for N in (768, 1024, 1536, 2048, 2304, 3072, 4096):
        bench(M, 768, N)
for K in (768, 1536, 3072):
    bench(6272, K, 768); bench(8192, K, 1024)
