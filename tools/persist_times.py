"""Where the one-launch loop's time goes: per-workgroup, per-phase clocks summed over the steps (debug build, -DND_PERSIST_TIMING).
   build:  python -c "from nested_diffusion_amd import build; build.build(out='tools/bin/libnd_hip_pt.so', defines=['ND_PERSIST_TIMING'])"
   run:    ND_LIB_PATH=tools/bin/libnd_hip_pt.so python tools/persist_times.py [K T B] [--skew us]"""
import argparse, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import synthetic, _lib
from nested_diffusion_amd.engine import EnsembleEngine
from nested_diffusion_amd.diffusion_utils import make_beta_schedule

ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="*", type=int)
ap.add_argument("--skew", type=float, default=0.0)
a = ap.parse_args()
K, T, B = (a.shape + [5, 100, 32][len(a.shape):])[:3]
D, H, F, Cc = 1024, 4096, 4096, 2
dev = torch.device("cuda", 0)
torch.manual_seed(0)
eng = EnsembleEngine(Cc, D, H, F, T, n_members=K, max_batch=B, device=dev)
lib = _lib.load()
buf = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
lib.nd_debug_set_persist_times.argtypes = [C.c_void_p]
assert lib.nd_debug_set_persist_times(buf.data_ptr()) == 0
for k in range(K):
    eng.load_member(k, synthetic.cond_model_state(D, H, F, Cc, T, seed=1000 + k, device=dev, denoiser=True))
betas = make_beta_schedule("linear", T, 1e-4, 0.02).to(dev)
alphas = 1 - betas
eng.set_schedule(alphas, torch.sqrt(1 - torch.cumprod(alphas, 0)))
eng.encode(torch.randn(B, D, device=dev))
yhat = torch.softmax(torch.randn(K, B, Cc, device=dev), -1)
noise = torch.randn(K, T, B, Cc, device=dev)
eng.set_loop_form(True, a.skew)
for _ in range(3):
    eng.sample(yhat, yhat, noise, mc=1, T=T)
torch.cuda.synchronize()
assert eng.loop_form() == "one_launch"
eng.persist_status()
names = ["head: eps + posterior", "head: h1", "wait 1", "lin2 loop", "lin2 epilogue", "wait 2", "lin3 loop", "lin3 epilogue", "wait 3"]
t = buf.cpu().reshape(256, 16)[: (256 // K) * min(K, int(os.environ.get("ND_PERSIST_ACTIVE", K))), :9].double() / 100.0 / T          # us per step
print(f"K={K} T={T} B={B} skew={a.skew} us: per-step time per phase, mean / min / max over the {t.shape[0]} workgroups (us)")
for k, n in enumerate(names):
    print(f"  {n:24s} {t[:, k].mean():7.2f} {t[:, k].min():7.2f} {t[:, k].max():7.2f}")
print(f"  {'sum':24s} {t.sum(1).mean():7.2f}")
wpm = 256 // K
for g in range(int(os.environ.get("ND_PERSIST_ACTIVE", K))):
    sl = t[g * wpm:(g + 1) * wpm]
    print(f"  member {g}: " + "  ".join(f"{sl[:, k].mean():6.2f}" for k in range(9)) + f"   | six-fragment workgroup: " + "  ".join(f"{sl[0, k]:6.2f}" for k in range(9)))
