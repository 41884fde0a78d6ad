#!/bin/bash
# Host-side sanitizer pass over the library's plan arithmetic, descriptor packing and argument checks.  CPU BUILD ONLY -- never on a GPU box
# (GPU AddressSanitizer is not available on this pool): the HOST pass of every translation unit is compiled with
# -fsanitize=address,undefined, the device pass is left alone (-fno-gpu-sanitize), and the `-m "not gpu"` tests that load the library
# (ABI surface, nd_skinny_plan / nd_step_plan / nd_b9 plans, argument checks, nd_last_error) run against it with the ASan runtime preloaded.
#   tools/sanitize_host.sh            -> profiles/r06_host_sanitizer.txt
set -o pipefail
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r06_host_sanitizer.txt}
RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
[ -n "$RT" ] || { echo "no ASan runtime next to hipcc's clang"; exit 1; }
python - <<'PY' || exit 1
from nested_diffusion_amd import build
print(build.build(out="tools/bin/libnd_hip_asan.so", extra_flags=["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g"],
                  link_flags=["-fsanitize=address,undefined", "-fno-gpu-sanitize"]))
PY
{
  echo "# host-side AddressSanitizer + UndefinedBehaviorSanitizer build of libnd_hip.so (device code not instrumented), $(date -u +%F)"
  echo "# runtime: $RT"
  echo "# command: ND_LIB_PATH=tools/bin/libnd_hip_asan.so LD_PRELOAD=\$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python -m pytest tests/test_abi_and_host.py tests/test_cli_host.py -q -m 'not gpu' -p no:cacheprovider"
  ND_LIB_PATH=$PWD/tools/bin/libnd_hip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest tests/test_abi_and_host.py tests/test_cli_host.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -40
  echo "# exit status of pytest: ${PIPESTATUS[0]}"
} | tee "$OUT"
