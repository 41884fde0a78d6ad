"""Per-workgroup clocks of the step GEMMs inside the REAL sampler graph (debug build of the library with -DND_WG_TIMING).
   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DND_WG_TIMING nested_diffusion_amd/csrc/*.hip -o tools/libnd_hip_dbg.so
   run:    ND_LIB_PATH=tools/libnd_hip_dbg.so python tools/wg_times.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import synthetic, _lib
from nested_diffusion_amd.engine import EnsembleEngine
from nested_diffusion_amd.diffusion_utils import make_beta_schedule

K, T, B = 5, 40, 32
D, H, F, Cc = 1024, 4096, 4096, 2
dev = torch.device("cuda", 0)
torch.manual_seed(0)
eng = EnsembleEngine(Cc, D, H, F, T, n_members=K, max_batch=B, device=dev)
lib = _lib.load()
buf = torch.zeros(3 * 8192 * 3, dtype=torch.int64, device=dev)
lib.nd_debug_set_wg_times.argtypes = [C.c_void_p]
assert lib.nd_debug_set_wg_times(buf.data_ptr()) == 0
for k in range(K):
    eng.load_member(k, synthetic.cond_model_state(D, H, F, Cc, T, seed=1000 + k, device=dev))
betas = make_beta_schedule("linear", T, 1e-4, 0.02).to(dev)
alphas = 1 - betas
eng.set_schedule(alphas, torch.sqrt(1 - torch.cumprod(alphas, 0)))
eng.encode(torch.randn(B, D, device=dev))
yhat = torch.softmax(torch.randn(K, B, Cc, device=dev), -1)
noise = torch.randn(K, T, B, Cc, device=dev)
for _ in range(3):
    eng.sample(yhat, yhat, noise, mc=1, T=T)
torch.cuda.synchronize()
h = buf.cpu().reshape(3, 2, 4096, 3)
for mode, name in ((0, "lin2 (MODE 0)"), (1, "lin3+lin4 (MODE 1)")):
    t = h[mode, 0, :256].double(); u = h[mode, 1, :256].double()
    t0 = t[:, 0].min()
    start, loop_end, end = (t[:, 0] - t0) / 100, (t[:, 1] - t0) / 100, (t[:, 2] - t0) / 100
    redw, act = (u[:, 0] - t0) / 100, (u[:, 1] - t0) / 100
    print(f"{name}: start max {start.max():.2f}  loop-end mean {loop_end.mean():.2f} max {loop_end.max():.2f}  "
          f"activated-loop mean {(act - loop_end).mean():.2f}  end-activated mean {(end - act).mean():.2f}  end max {end.max():.2f} us")
    heavy = [0, 51, 102, 153, 204]
    print("   6-fragment workgroups end at", [round(float(end[i]), 1) for i in heavy])
    for g in range(K):       # per member (51 workgroups each): members 0..2 of lin2 read Infinity-Cache resident weights
        sl = slice(51 * g, 51 * g + 51)
        print(f"   member {g}: loop-end mean {loop_end[sl].mean():.2f} max {loop_end[sl].max():.2f}  end max {end[sl].max():.2f} us")
    order = torch.argsort(end, descending=True)[:8]
    print("   latest workgroups:", [(int(i), round(float(loop_end[i]), 1), round(float(end[i]), 1)) for i in order])
