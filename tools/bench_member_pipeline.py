"""Does running the members' step chains as INDEPENDENT graphs on several streams fill the holes of one chain (step head, dispatch
gaps, the slowest workgroup's tail) with another chain's weight streaming?  Sampler only, K = 5, T = 100, B = 32, config dims.
   python3 tools/bench_member_pipeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import synthetic
from nested_diffusion_amd._lib import check, ptr
from nested_diffusion_amd.engine import EnsembleEngine
from nested_diffusion_amd.diffusion_utils import make_beta_schedule

K, T, B, mc = 5, 100, 32, 1
D, H, F, C = 1024, 4096, 4096, 2
dev = torch.device("cuda", 0)
torch.manual_seed(0)
eng = EnsembleEngine(C, D, H, F, T, n_members=K, max_batch=B, max_rows=B * mc, device=dev)
for k in range(K):
    eng.load_member(k, synthetic.cond_model_state(D, H, F, C, T, seed=1000 + k, device=dev))
betas = make_beta_schedule("linear", T, 1e-4, 0.02).to(dev)
alphas = 1 - betas
eng.set_schedule(alphas, torch.sqrt(1 - torch.cumprod(alphas, 0)))
eng.encode(torch.randn(B, D, device=dev))
yhat = torch.softmax(torch.randn(K, B, C, device=dev), -1)
noise = torch.randn(K, T, B * mc, C, device=dev)
y0 = torch.empty(K, B * mc, C, device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(K)]


def run(parts):
    m0 = 0
    main = torch.cuda.current_stream(dev)
    for i, nm in enumerate(parts):
        st = streams[i]
        st.wait_stream(main)
        check(eng.lib.nd_sample(eng.h, m0, nm, ptr(yhat[m0:]), ptr(yhat[m0:]), ptr(noise[m0:]), ptr(y0[m0:]), None, B, mc, T, 1,
                                st.cuda_stream), "nd_sample")
        m0 += nm
    for i in range(len(parts)):
        main.wait_stream(streams[i])


ref = None
for parts in ([5], [3, 2], [2, 3], [2, 2, 1], [1, 1, 1, 1, 1], [4, 1]):
    for _ in range(2):
        run(parts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        run(parts)
    e1.record(); torch.cuda.synchronize()
    cs = float(y0.double().sum())
    ref = cs if ref is None else ref
    print(f"member groups {parts}: sampler {e0.elapsed_time(e1) / reps:.3f} ms per batch; checksum {cs:.9f} (delta {cs - ref:.2e})", flush=True)
