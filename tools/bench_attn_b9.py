"""The two fp32 attention paths of a ViT block side by side on one box (B x 196 tokens x 12 heads):
   f32   qkv Linear (nd_gemm_split, fp32 output) + k_attention_ring (v_mfma_f32_16x16x4_f32), result as the proj GEMM's input image
   b9    qkv Linear writing the attention's operand images (nd_gemm_split_qkv) + k_attention_b9 (bf16 matrix pipe, exact fp32 products)
each kernel launched back to back, and the pair as it runs inside a block.   python tools/bench_attn_b9.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import ops

B, N, heads = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 196, 12
E = heads * 64
g = torch.Generator().manual_seed(0)
x = torch.randn(B * N, E, generator=g).cuda()
w = (torch.randn(3 * E, E, generator=g) / E ** 0.5).cuda()
b = torch.randn(3 * E, generator=g).cuda()
xs, ws = ops.split_rows(x), ops.split_rows(w)


def timed(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


qkv = ops.gemm_split(xs, ws, b)
img = ops.gemm_split_qkv(xs, ws, b, B, N, heads)
fl = B * heads * 2 * (2 * N * N * 64)
t = {"qkv GEMM, fp32 output": timed(lambda: ops.gemm_split(xs, ws, b)),
     "qkv GEMM, attention images": timed(lambda: ops.gemm_split_qkv(xs, ws, b, B, N, heads)),
     "k_attention_ring (f32 MFMA)": timed(lambda: ops.attention_split(qkv, B, N, heads)),
     "k_attention_b9 (bf16 x 9)": timed(lambda: ops.attention_images(img, B, N, heads, want_split=True)),
     "pair f32": timed(lambda: ops.attention_split(ops.gemm_split(xs, ws, b), B, N, heads)),
     "pair b9": timed(lambda: ops.attention_images(ops.gemm_split_qkv(xs, ws, b, B, N, heads), B, N, heads, want_split=True))}
for k, v in t.items():
    extra = ""
    if k.startswith("k_attention_ring"): extra = f"  {fl / v / 1e6:6.1f} TFLOP/s = {fl / v / 1e6 / 157.3:.2f} of the f32-MFMA peak"
    if k.startswith("k_attention_b9"): extra = f"  {fl / v / 1e6:6.1f} TFLOP/s of useful fp32 flop = {fl / v / 1e6 / (2500 / 9):.2f} of 2500 / 9"
    print(f"B={B}  {k:32s} {v:8.1f} us{extra}", flush=True)
d = (ops.join_rows(ops.attention_images(img, B, N, heads, want_split=True)) - ops.join_rows(ops.attention_split(qkv, B, N, heads))).abs().max().item()
print(f"max |b9 - f32| = {d:.2e}")
