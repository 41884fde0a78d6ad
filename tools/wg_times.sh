# Per-workgroup clocks of the step kernels: a debug variant of the WHOLE library (every source of nested_diffusion_amd.build.SOURCES,
# -DND_WG_TIMING) next to the product one, loaded through ND_LIB_PATH.  Build errors are shown, not swallowed.
cd $GRAFT_REPO_ROOT
python3 -c "from nested_diffusion_amd import build; print(build.build(out='/tmp/libnd_hip_dbg.so', defines=['ND_WG_TIMING']))" || exit 1
for s in 0 4; do echo "== ND_TAIL_SPLIT=$s"; ND_TAIL_SPLIT=$s ND_LIB_PATH=/tmp/libnd_hip_dbg.so python3 tools/wg_times.py 2>&1 | grep -v amdgpu.ids; done
