// nd_skinny_m0.hip -- the k_skinny<..., MODE 0, ...> family: a ConditionalLinear block / a Linear layer with its folded scale, shift and activation.
// One translation unit per MODE (the three compile side by side; nd_common.hpp explains the kernel and the launch plan).  gfx950 only.
#define ND_SKINNY_MODE 0
#include "nd_common.hpp"

SkinnyLaunch nd_skinny_launch_m0(int K, int N, int M, int nm, int half) { return nd_skinny_launch_impl<0>(K, N, M, nm, half); }
#ifdef ND_WG_TIMING
// debug builds only (tools/wg_times.py): this translation unit's copy of the clock buffer pointer
int nd_debug_set_wg_times_m0(void* dev_ptr) { return hipMemcpyToSymbol(HIP_SYMBOL(nd_dbg_times), &dev_ptr, sizeof dev_ptr) == hipSuccess ? 0 : -1; }
#endif
