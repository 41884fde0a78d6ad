import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import ops
B, N, heads = 32, 196, 12
qkv = torch.randn(B * N, 3 * heads * 64, device="cuda")
for _ in range(3): ops.attention(qkv, B, N, heads)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): ops.attention(qkv, B, N, heads)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
fl = B * heads * 2 * (2 * N * N * 64)
print(f"attention B={B} N={N} heads={heads}: {us:.1f} us, {fl/us/1e6:.1f} TFLOP/s = {100*fl/us/1e6/157.3:.1f}% of f32 MFMA peak")
