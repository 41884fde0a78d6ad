"""Conditioner breakdown at config dims: ViT prefix blocks vs mapping-MLP layers.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import ops, synthetic
from nested_diffusion_amd.mapping import Classifier, VisionTransformer, GuidingConditioner

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

dev = "cuda"
B = 32
x = torch.randn(B, 150528, device=dev)
for K_, N_ in ((150528, 4096), (4096, 2048), (2048, 128), (128, 2)):
    w = ops.PackedWeight(torch.randn(N_, K_, device=dev) / K_ ** 0.5)
    b = torch.randn(N_, device=dev)
    xi = torch.randn(B, K_, device=dev)
    ms = timed(lambda: ops.linear(xi, w, b, act="relu"))
    print(f"linear M={B} K={K_} N={N_}: {ms*1e3:8.1f} us  {N_*K_*4/ms/1e6:7.1f} GB/s", flush=True)
vit = VisionTransformer(synthetic.vit_state(seed=7, device=dev), 12, dev)
mlps = [Classifier(synthetic.classifier_state(196 * 768, seed=2000 + k, device=dev), dev) for k in range(5)]
cond = GuidingConditioner(vit, mlps)
img = synthetic.images(B, device=dev)
print(f"conditioner total: {timed(lambda: cond.compute_guiding_prediction(img, include_full_vit=False)):.3f} ms", flush=True)
tok = torch.randn(B * 196, 768, device=dev)
print(f"patch embed: {timed(lambda: vit.patch_embed(img))*1e3:.1f} us")
print(f"one ViT block: {timed(lambda: vit.block(0, tok, B))*1e3:.1f} us")
print(f"one classifier: {timed(lambda: mlps[0](tok.reshape(B, 196, 768)))*1e3:.1f} us")
