"""Does the conditioner of batch i+1 hide under the sampler of batch i?  Two streams, config dims.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = bench.ns(timesteps=100, members=5, batch=32, mc=1, cpu_baseline=False)
dev = torch.device("cuda", 0)
runner, cfg, _ = bench.build_runner(args, dev)
from nested_diffusion_amd import synthetic
eng = runner.engine
B, K, T, C = 32, 5, 100, 2
images = synthetic.images(B, device=dev)
flat = torch.flatten(images, 1)
yhat = torch.softmax(torch.randn(K, B, C, device=dev), -1)
noise = torch.randn(K, T, B, C, device=dev)
eng.encode(flat)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def samp():
    with torch.cuda.stream(sA):
        eng.sample(yhat, yhat, noise, mc=1, T=T)
def cond():
    with torch.cuda.stream(sB):
        runner.cond_pred_model.compute_guiding_prediction(images, include_full_vit=False)
def enc():
    with torch.cuda.stream(sB):
        eng.encode(flat)

def wall(fn, reps=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

print(f"sampler alone      {wall(samp):7.2f} ms")
print(f"conditioner alone  {wall(cond):7.2f} ms")
print(f"both concurrently  {wall(lambda: (samp(), cond())):7.2f} ms   (sum of the two alone = sequential)")
