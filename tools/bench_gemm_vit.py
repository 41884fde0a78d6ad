"""The five GEMM shapes of one ViT-B/16 block at B = 32 (M = 6272 token rows), with their epilogues, three ways:
     nd_gemm_split       exact fp32 products on the bf16 matrix pipe, operands pre-split (frag32b3): the product path
     nd_gemm_bias_act    the f32-input MFMA kernel k_gemm_nt (rounds 1-3)
     torch.mm            rocBLAS / hipBLASLt sgemm, bare product without epilogue: a yardstick, never on the product path
   python tools/bench_gemm_vit.py [B]      -> us and TFLOP/s of useful fp32 flop per shape (back-to-back launches of ONE shape: the
   matrix pipe is busy throughout, so the clock settles lower than inside the conditioner, where GEMMs alternate with LayerNorm /
   attention), block totals, relative error against an fp64 product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nested_diffusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = B * 196
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [("patch_embed", 768, 768, None, False), ("qkv", 768, 2304, None, False), ("proj", 768, 768, None, True),
          ("fc1", 768, 3072, "gelu", False), ("fc2", 3072, 768, None, True)]
only = os.environ.get("ND_GEMM_ONLY")            # one shape only (tools/pmc_gemm.sh: PMC passes per shape)
tot = tot32 = 0.0
for name, K, N, act, res in shapes:
    if only and name != only:
        continue
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g) if res else None
    reps = 100
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xs, ws = ops.split_rows(x), ops.split_rows(w)
    for _ in range(20):
        y = ops.gemm_split(xs, ws, b, act=act, residual=r)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        y = ops.gemm_split(xs, ws, b, act=act, residual=r)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    for _ in range(20):                      # ops.gemm_bias_act on fp32 weights IS the f32-input-MFMA kernel (k_gemm_nt): no switch involved
        y32 = ops.gemm_bias_act(x, w, b, act=act, residual=r)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        y32 = ops.gemm_bias_act(x, w, b, act=act, residual=r)
    e1.record(); torch.cuda.synchronize()
    us32 = e0.elapsed_time(e1) / reps * 1e3
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r.double()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    err32 = float((y32.double() - ref).abs().max() / ref.abs().max())
    if name != "patch_embed":
        tot += us; tot32 += us32
    # library yardstick (not used by the product): the same product through torch.mm = rocBLAS / hipBLASLt sgemm, no epilogue
    wt = w.t().contiguous()
    for _ in range(3):
        torch.mm(x, wt)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        torch.mm(x, wt)
    e1.record(); torch.cuda.synchronize()
    us_lib = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:12s} M={M} K={K:4d} N={N:4d}: bf16x9 {us:7.1f} us {2 * M * K * N / us / 1e6:6.1f} TFLOP/s err {err:.1e} | f32 MFMA {us32:7.1f} us "
          f"{2 * M * K * N / us32 / 1e6:6.1f} TFLOP/s err {err32:.1e} | torch.mm fp32, bare product {us_lib:7.1f} us {2 * M * K * N / us_lib / 1e6:6.1f} TFLOP/s",
          flush=True)
print(f"block GEMMs (qkv+proj+fc1+fc2): bf16x9 {tot:.1f} us, f32 MFMA {tot32:.1f} us ({tot32 / tot:.2f}x)")
