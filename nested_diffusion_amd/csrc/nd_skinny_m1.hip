// nd_skinny_m1.hip -- the k_skinny<..., MODE 1, ...> family: the same block followed by the lin4 projection (eps partials per 16-column fragment).
// One translation unit per MODE (the three compile side by side; nd_common.hpp explains the kernel and the launch plan).  gfx950 only.
#define ND_SKINNY_MODE 1
#include "nd_common.hpp"

SkinnyLaunch nd_skinny_launch_m1(int K, int N, int M, int nm, int half) { return nd_skinny_launch_impl<1>(K, N, M, nm, half); }
#ifdef ND_WG_TIMING
// debug builds only (tools/wg_times.py): this translation unit's copy of the clock buffer pointer
int nd_debug_set_wg_times_m1(void* dev_ptr) { return hipMemcpyToSymbol(HIP_SYMBOL(nd_dbg_times), &dev_ptr, sizeof dev_ptr) == hipSuccess ? 0 : -1; }
#endif
