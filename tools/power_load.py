"""Keeps ONE kernel form running back to back for a while (tools/power_probe.sh samples rocm-smi meanwhile).
   python tools/power_load.py {b9|f32|b9_cond|stream} [seconds]      GPU only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nested_diffusion_amd import ops, _lib

which, secs = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
g = torch.Generator().manual_seed(0)
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
if which in ("b9", "f32"):
    M, K, N = 6272, 768, 2304
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); out = torch.empty(M, N, device="cuda")
    xs, ws = ops.split_rows(x), ops.split_rows(w)
    if which == "b9":
        fn = lambda: lib.nd_gemm_split(xs.data.data_ptr(), ws.data.data_ptr(), None, None, out.data_ptr(), None, M, K, N, 0, None, 0, st)
    else:
        fn = lambda: lib.nd_gemm_bias_act(x.data_ptr(), w.data_ptr(), None, None, out.data_ptr(), M, K, N, 0, 0, None, 0, st)
elif which == "b9_cond":
    M, K, N = 3200, 4096, 4096          # 5 members x 640 rows worth of tiles through the wide split-GEMM
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); out = torch.empty(M, N, device="cuda")
    xs, ws = ops.split_rows(x), ops.split_rows(w)
    fn = lambda: lib.nd_gemm_split(xs.data.data_ptr(), ws.data.data_ptr(), None, None, out.data_ptr(), None, M, K, N, 0, None, 0, st)
else:
    big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(big)
    fn = lambda: dst.copy_(big)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50): fn()
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print(f"{which}: {n} launches in {dt:.1f} s = {dt / n * 1e6:.1f} us per launch")
