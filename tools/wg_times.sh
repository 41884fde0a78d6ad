cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DND_WG_TIMING nested_diffusion_amd/csrc/nd_sampler.hip nested_diffusion_amd/csrc/nd_ops.hip nested_diffusion_amd/csrc/nd_vit.hip nested_diffusion_amd/csrc/nd_image.hip nested_diffusion_amd/csrc/nd_cond_gemm.hip nested_diffusion_amd/csrc/nd_attention.hip nested_diffusion_amd/csrc/nd_gemm_f32.hip -o /tmp/libnd_hip_dbg.so 2>/dev/null
for s in 0 4; do echo "== ND_TAIL_SPLIT=$s"; ND_TAIL_SPLIT=$s ND_LIB_PATH=/tmp/libnd_hip_dbg.so python3 tools/wg_times.py 2>&1 | grep -v amdgpu.ids; done
