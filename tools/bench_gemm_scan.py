import sys, os
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from nested_diffusion_amd import ops
def bench(M, K, N, act=None, reps=10):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    for _ in range(3): ops.gemm_bias_act(x, w, b, act=act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ops.gemm_bias_act(x, w, b, act=act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tiles = ((M + 127) // 128) * ((N + 63) // 64)
    print(f"M={M:6d} K={K:5d} N={N:5d}: tiles {tiles:6d} ({tiles/1024:.2f} x 1024 slots) {us:8.1f} us  {2*M*K*N/us/1e6:7.1f} TFLOP/s", flush=True)
for M in (1024, 2048, 4096, 6272, 8192, 16384, 32768, 65536):
    bench(M, 768, 2304)
for M in (4096, 6272, 16384, 65536):
    bench(M, 3072, 768)
