// nd_skinny_m2.hip -- the k_skinny<..., MODE 2, ...> family: split-K: raw partial sums per k-slab, finished by k_splitk_epilogue (encoder_x.0, mapping linear1).
// One translation unit per MODE (the three compile side by side; nd_common.hpp explains the kernel and the launch plan).  gfx950 only.
#define ND_SKINNY_MODE 2
#include "nd_common.hpp"

SkinnyLaunch nd_skinny_launch_m2(int K, int N, int M, int nm, int half) { return nd_skinny_launch_impl<2>(K, N, M, nm, half); }
#ifdef ND_WG_TIMING
// debug builds only (tools/wg_times.py): this translation unit's copy of the clock buffer pointer
int nd_debug_set_wg_times_m2(void* dev_ptr) { return hipMemcpyToSymbol(HIP_SYMBOL(nd_dbg_times), &dev_ptr, sizeof dev_ptr) == hipSuccess ? 0 : -1; }
#endif
