"""Per-phase clocks of k_attention_ring (debug build with -DND_ATT_STAMPS).  GPU box:
     python tools/att_stamps.py [B ...]
builds a private copy of the library with the stamps compiled in, runs the attention at each B and prints, per phase, the
mean over the active waves of the time spent (us at 100 MHz s_memtime ticks -> shader cycles / clock) ."""
import os, subprocess, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "nested_diffusion_amd", "csrc")
lib = "/tmp/libnd_hip_stamps.so"
srcs = ["nd_sampler.hip", "nd_ops.hip", "nd_vit.hip", "nd_image.hip", "nd_cond_gemm.hip", "nd_attention.hip", "nd_gemm_f32.hip",
        "nd_conditioner.hip", "nd_rng.hip"]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DND_ATT_STAMPS", "-mllvm", "-amdgpu-mfma-vgpr-form", "-o", lib] + [os.path.join(csrc, s) for s in srcs]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
os.environ["ND_LIB_PATH"] = lib            # read by nested_diffusion_amd._lib at import
import torch
from nested_diffusion_amd import _lib, ops
h = _lib.load()
h.nd_debug_set_att_stamps.argtypes = [C.c_void_p]
N, heads = 196, 12
names = ["start->K0 landed", "S tile0", "sync+DMA", "S tile1", "sync+DMA", "S tile2", "sync+DMA+softmax", "PV tile0", "sync+DMA", "PV tile1", "sync", "PV tile2", "store"]
for B in [int(v) for v in sys.argv[1:]] or [2, 16, 32]:
    qkv = torch.randn(B * N, 3 * heads * 64, device="cuda")
    nwg = B * heads * 4
    st = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device="cuda")
    assert h.nd_debug_set_att_stamps(C.c_void_p(st.data_ptr())) == 0
    for _ in range(3):
        ops.attention(qkv, B, N, heads, "f32")
    torch.cuda.synchronize()
    t = st.cpu().reshape(nwg, 4, 16).double()
    act = t[:, :, 2] - t[:, :, 1] > 50            # waves that computed something in S tile0
    d = (t[:, :, 1:14] - t[:, :, 0:13])
    span = (t[:, :, 13] - t[:, :, 0])
    print(f"B={B}: {nwg} workgroups; wave lifetime mean {span[act].mean():.0f} ticks, kernel span {(t[:, :, 13].max() - t[:, :, 0].min()):.0f} ticks")
    for i, n in enumerate(names):
        print(f"    {n:22s} {d[:, :, i][act].mean():9.0f}")
    # when do workgroups start (first tile landed) and end, relative to the first start?  (cycles; by dispatch order)
    t1, t13 = t[:, 0, 1], t[:, 0, 13]
    z = t1.min()
    q = torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)
    for lo, hi in ((0, 768), (768, 1024), (1024, nwg)):
        if lo >= nwg:
            break
        a, b = t1[lo:min(hi, nwg)] - z, t13[lo:min(hi, nwg)] - z
        print(f"    workgroups {lo:4d}..{min(hi, nwg) - 1:4d}: first tile ready at {[int(v) for v in torch.quantile(a, q)]}, end at {[int(v) for v in torch.quantile(b, q)]} (min, 10 %, median, 90 %, max)")
