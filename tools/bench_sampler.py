"""Sampler-only timing (K members x T steps in one hipGraph) with the in-graph kernel probes.  GPU only.
   [ND_DTYPE=f16] [ND_BENCH_F=4080] python tools/bench_sampler.py [K T B mc]
   ND_BENCH_F: feature dim (4080 = 255 fragments per member = exactly 5 per workgroup at K = 5: what perfect balance is worth)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import synthetic
from nested_diffusion_amd.engine import EnsembleEngine
from nested_diffusion_amd.diffusion_utils import make_beta_schedule

K, T, B, mc = (int(v) for v in (sys.argv[1:5] + ["5", "100", "32", "1"][len(sys.argv) - 1:]))
D, H, F, C = 1024, 4096, int(os.environ.get("ND_BENCH_F", "4096")), 2     # the step loop never touches data_dim: a small encoder keeps set-up short
dev = torch.device("cuda", 0)
torch.manual_seed(0)
DT = os.environ.get("ND_DTYPE", "f32")
eng = EnsembleEngine(C, D, H, F, T, n_members=K, max_batch=B, max_rows=B * mc, device=dev, dtype=DT)
for k in range(K):
    eng.load_member(k, synthetic.cond_model_state(D, H, F, C, T, seed=1000 + k, device=dev))
betas = make_beta_schedule("linear", T, 1e-4, 0.02).to(dev)
alphas = 1 - betas
eng.set_schedule(alphas, torch.sqrt(1 - torch.cumprod(alphas, 0)))
eng.encode(torch.randn(B, D, device=dev))
yhat = torch.softmax(torch.randn(K, B, C, device=dev), -1)
noise = torch.randn(K, T, B * mc, C, device=dev)
eng.set_profiling(True)
for _ in range(2):
    y = eng.sample(yhat, yhat, noise, mc=mc, T=T)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get("ND_BENCH_REPS", "5"))     # a long loop (e.g. 80) gives tools/power_probe_sampler.sh time to sample
e0.record()
for _ in range(reps):
    y = eng.sample(yhat, yhat, noise, mc=mc, T=T)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
head, pair, rec, n = eng.profile_read()
print(f"dtype={DT} K={K} T={T} B={B} mc={mc}: sampler {ms:.3f} ms = {ms*1e3/T:.1f} us/step; "
      f"head {head - rec:.1f} us, step block {(pair - rec) / 2:.1f} us per launch (record-node overhead {rec:.1f} us, calibrated in the graph, subtracted from the "
      f"two-launch interval; {n} probed pairs of steps); "
      f"checksum {float(y.double().sum()):.9f}")
