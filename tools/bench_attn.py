"""Attention kernel timing and MFMA utilisation: python tools/bench_attn.py [f32|f16]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import ops
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
B, N, heads = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 196, 12
qkv = torch.randn(B * N, 3 * heads * 64, device="cuda")
for _ in range(3): ops.attention(qkv, B, N, heads, dt)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): ops.attention(qkv, B, N, heads, dt)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
fl = B * heads * 2 * (2 * N * N * 64)
peak, name = (157.3, "f32 MFMA") if dt == "f32" else (2500.0, "dense f16 MFMA")
print(f"attention {dt} B={B} N={N} heads={heads}: {us:.1f} us, {fl/us/1e6:.1f} TFLOP/s = {100*fl/us/1e6/peak:.1f}% of the {name} peak")
