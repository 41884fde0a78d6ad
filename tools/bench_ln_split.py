"""LayerNorm with a frag32b3 (split) output against the fp32 LayerNorm and the stand-alone split pass, [6272, 768].  GPU only."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nested_diffusion_amd import ops, _lib
x = torch.randn(6272, 768, device="cuda"); w = torch.randn(768, device="cuda"); b = torch.randn(768, device="cuda")
def timed(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
out = ops.SplitMatrix(6272, 768, "cuda")
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
print("layernorm f32      %.1f us" % timed(lambda: ops.layernorm(x, w, b, 1e-6)))
print("layernorm split    %.1f us" % timed(lambda: lib.nd_layernorm_split(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data.data_ptr(), 6272, 768, 1e-6, st)))
print("layernorm split (rows=63: direct form) x100 rows-equivalent: %.1f us" % timed(lambda: lib.nd_layernorm_split(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data.data_ptr(), 63, 768, 1e-6, st)))
print("split_rows         %.1f us" % timed(lambda: ops.split_rows(x, out)))
