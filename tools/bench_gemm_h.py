"""fp16-operand GEMM throughput on the ViT shapes (TFLOP/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import ops
def bench(M, K, N, half, reps=20):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    if half: w = w.half()
    for _ in range(3): ops.gemm_bias_act(x, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ops.gemm_bias_act(x, w, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{'f16' if half else 'f32'} M={M:6d} K={K:5d} N={N:5d}: {us:8.1f} us  {2*M*K*N/us/1e6:7.1f} TFLOP/s", flush=True)
bench(4096, 4096, 4096, True)
for shape in [(6272, 768, 2304), (6272, 768, 768), (6272, 768, 3072), (6272, 3072, 768), (8192, 4096, 4096)]:
    bench(*shape, True)
