"""Builds libnd_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnd_hip.so")
SOURCES = ["nd_skinny_m0.hip", "nd_skinny_m1.hip", "nd_skinny_m2.hip", "nd_sampler.hip", "nd_ops.hip", "nd_vit.hip", "nd_image.hip", "nd_cond_gemm.hip", "nd_attention.hip", "nd_gemm_f32.hip",
           "nd_conditioner.hip", "nd_rng.hip", "nd_gemm_b9.hip", "nd_persist.hip"]
# per-file flags: the large-M tile kernel keeps its accumulators in VGPRs (see nd_cond_gemm.hpp)
EXTRA_FLAGS = {"nd_cond_gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "nd_attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "nd_gemm_f32.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "nd_gemm_b9.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
HEADERS = [os.path.join(CSRC, "nd_common.hpp"), os.path.join(CSRC, "nd_cond_gemm.hpp"), os.path.join(CSRC, "nd_rng.hpp"), os.path.join(CSRC, "nd_b9.hpp"), os.path.join(CSRC, "nd_step.hpp"), os.path.join(CSRC, "nd_persist.hpp"), os.path.join(os.path.dirname(HERE), "include", "nested_diffusion.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Werror=inline-asm"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnd_hip.so cannot be built")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, out: str = None, defines=(), extra_flags=(), link_flags=()) -> str:
    """out / defines: an experimental variant next to the product library (e.g. defines=["ND_SKINNY_B9=1"], out=".../libnd_hip_x.so";
    loaded with ND_LIB_PATH=<that file>).  extra_flags / link_flags: more compiler / linker flags for such a variant (the host-side
    sanitizer build of tools/sanitize_host.sh: -fsanitize=address,undefined -fno-gpu-sanitize)."""
    if out is None and not defines and not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build" if out is None else "build_" + os.path.basename(out).replace(".", "_"))
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *["-D" + d for d in defines], *extra_flags, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr:
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, SOURCES))
    target = out or LIB
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *link_flags, "-o", target, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
