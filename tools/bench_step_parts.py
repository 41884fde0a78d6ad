"""Where one bench step (runner.predict_batch at the headline config) spends its time beyond the three big stages.
   python3 tools/bench_step_parts.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nested_diffusion_amd import ops, synthetic

args = types.SimpleNamespace(batch=32, members=5, timesteps=100, mc=1, dtype="f32")
dev = torch.device("cuda:0")
runner, cfg = bench.build_runner(args, dev)
eng = runner.engine
B, K, T, C = 32, 5, 100, 2
images = synthetic.images(B, seed=1234, device=dev)
flat = torch.flatten(images, 1)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


t_all, out = timed(lambda: runner.predict_batch(images))
t_cond, logits = timed(lambda: runner.cond_pred_model.compute_guiding_prediction_py(images, include_full_vit=False, side_work=lambda: eng.encode(flat)))
t_soft, yhat = timed(lambda: torch.stack([ops.softmax_rows(logits[i]) for i in runner.members]))
t_rand, noise = timed(lambda: torch.randn(K, T, B, C, device=dev))
t_samp, y0 = timed(lambda: eng.sample(yhat, yhat, noise, mc=1, T=T))
buf = eng.static_buffers(K, B, 1, T, False)
from nested_diffusion_amd._lib import check, ptr
t_graph, _ = timed(lambda: check(eng.lib.nd_sample(eng.h, 0, K, ptr(buf["yhat"]), ptr(buf["ymean"]), ptr(buf["noise"]), ptr(buf["y0"]), None, B, 1, T, 1, eng._stream()), "nd_sample"))
t_agg, _ = timed(lambda: ops.aggregate(y0.reshape(K, B, C), runner.temperature, return_probs=True))
print(f"predict_batch {t_all:.3f} ms = conditioner+hoist {t_cond:.3f} + softmax x{K} {t_soft:.3f} + randn {t_rand:.3f} + sample {t_samp:.3f} "
      f"(graph replay alone {t_graph:.3f}) + aggregate {t_agg:.3f}  [sum {t_cond + t_soft + t_rand + t_samp + t_agg:.3f}]")
