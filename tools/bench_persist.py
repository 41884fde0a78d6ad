"""The T-step loop as ONE launch (csrc/nd_persist.hip) against the hipGraph of per-step kernels, same process, alternating.  GPU only.
   python tools/bench_persist.py [K T B mc] [--skews 0,10,20,30] [--reps 5] [--rounds 3]
Prints, per form, the sampler time per loop and per step; asserts the two forms return the same bits."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import synthetic
from nested_diffusion_amd.engine import EnsembleEngine
from nested_diffusion_amd.diffusion_utils import make_beta_schedule

ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="*", type=int)
ap.add_argument("--skews", default="0,10,20,30,40")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--F", type=int, default=int(os.environ.get("ND_BENCH_F", "4096")))
ap.add_argument("--C", type=int, default=2)
a = ap.parse_args()
K, T, B, mc = (a.shape + [5, 100, 32, 1][len(a.shape):])[:4]
D, H, F, C = 1024, a.F, a.F, a.C
dev = torch.device("cuda", 0)
torch.manual_seed(0)
eng = EnsembleEngine(C, D, H, F, T, n_members=K, max_batch=B, max_rows=B * mc, device=dev)
for k in range(K):
    eng.load_member(k, synthetic.cond_model_state(D, H, F, C, T, seed=1000 + k, device=dev, denoiser=True))
betas = make_beta_schedule("linear", T, 1e-4, 0.02).to(dev)
alphas = 1 - betas
eng.set_schedule(alphas, torch.sqrt(1 - torch.cumprod(alphas, 0)))
eng.encode(torch.randn(B, D, device=dev))
yhat = torch.softmax(torch.randn(K, B, C, device=dev), -1)
noise = torch.randn(K, T, B * mc, C, device=dev)


def timed(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y = eng.sample(yhat, yhat, noise, mc=mc, T=T)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        y = eng.sample(yhat, yhat, noise, mc=mc, T=T)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, y


eng.set_loop_form(False)
t_g, y_g = timed(2)
assert eng.loop_form() == "graph_nodes"
seq_g = eng.sample(yhat, yhat, noise, mc=mc, T=T, return_seq=True)
print(f"K={K} T={T} B={B} mc={mc} F={F} C={C}")
print(f"graph of per-step kernels: {t_g:.3f} ms per loop = {t_g * 1e3 / T:.2f} us per step")
best = {}
for sk in [float(v) for v in a.skews.split(",")]:
    eng.set_loop_form(True, sk)
    t_p, y_p = timed(2)
    form = eng.loop_form()
    eng.persist_status()
    same = torch.equal(y_p, y_g)
    d = (y_p - y_g).abs().max().item()
    print(f"one launch, skew {sk:5.1f} us: {t_p:.3f} ms per loop = {t_p * 1e3 / T:.2f} us per step  ({t_p / t_g:.3f} x graph)  form={form} "
          f"bit-identical={same} max|dy0|={d:.3e}")
    if form != "one_launch":
        print("  (the plan kept the per-step form for this shape)")
        break
    if not same:
        seq_p = eng.sample(yhat, yhat, noise, mc=mc, T=T, return_seq=True)
        dd = (seq_p - seq_g).abs().amax(dim=(0, 2, 3))
        first = int((dd > 0).nonzero()[0]) if (dd > 0).any() else -1
        print(f"  first differing state index {first} of {T}; per-state max diff (first 12): {[float(x) for x in dd[:12]]}")
    best[sk] = t_p
# alternating rounds at the best skew
if best:
    sk = min(best, key=best.get)
    print(f"alternating rounds, skew {sk} us:")
    for r in range(a.rounds):
        eng.set_loop_form(False); tg, yg = timed(a.reps)
        eng.set_loop_form(True, sk); tp, yp = timed(a.reps)
        eng.persist_status()
        print(f"  round {r}: graph {tg:.3f} ms ({tg * 1e3 / T:.2f} us/step) | one launch {tp:.3f} ms ({tp * 1e3 / T:.2f} us/step) = {tp / tg:.3f} x; "
              f"bit-identical={torch.equal(yp, yg)} (graph vs first graph run: {(yg - y_g).abs().max().item():.3e}, one launch vs it: {(yp - y_g).abs().max().item():.3e})")
