"""Conditioner of the headline config (ViT-B/16 prefix: MFMA-bound; encoder hoist + mapping MLPs: HBM-bound): each half alone,
one after the other on one stream, and side by side on two streams.  usage: python3 tools/bench_cond_overlap.py [B]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = types.SimpleNamespace(batch=B, members=5, timesteps=100, mc=1, dtype="f32")
dev = torch.device("cuda:0")
runner, cfg = bench.build_runner(args, dev)
cond, eng = runner.cond_pred_model, runner.engine
x = torch.rand(B, 3, 224, 224, device=dev)
xf = torch.flatten(x, 1)
side = torch.cuda.Stream(dev)


def vit_only():
    tok = cond.vit.patch_embed(x)
    toks = []
    for i in range(5):
        tok = cond.vit.block(i, tok, B)
        toks.append(tok)
    return toks


toks = vit_only()


def side_only():
    eng.encode(xf)
    return [cond.mlps[i](toks[i]) for i in range(5)]


def serial():
    return cond.compute_guiding_prediction_py(x, include_full_vit=False, side_work=lambda: eng.encode(xf))


def two_streams():
    return cond.compute_guiding_prediction_py(x, include_full_vit=False, side_stream=side, side_work=lambda: eng.encode(xf))


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, fn in (("ViT prefix alone", vit_only), ("encoder hoist + mapping MLPs alone", side_only), ("one stream", serial),
                 ("two streams", two_streams)):
    print(f"{name:40s} {timeit(fn):7.3f} ms", flush=True)
