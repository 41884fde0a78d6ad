// nd_cond_gemm.hpp -- large-M form of the ConditionalLinear blocks (gfx950 / CDNA4 only).
//
// The reference's default operating point is mc_trials = 20 Monte-Carlo trials (classification_train_separately.py:770-771)
// x batch_size 70 (configs/chest_x_ray.yml:66): M = B*mc = 1400 rows per member go through lin2 / lin3 at every step
// (latent_model.py:178-184).  At that M the blocks are compute-bound ([M,4096] x [4096,4096] per member: arithmetic
// intensity M/2 flop per weight byte), and the weight-streaming kernel of nd_common.hpp -- 64 rows per workgroup pass --
// would re-stream the 67 MB layer ceil(M/64) times.  This is the LDS-tiled kernel for that regime, on the SAME operands
// (frag16-packed weights and activations, same folded gain/shift tables, same outputs), so nothing else changes.
//
// Tile 128 (W rows = output columns) x 128 (activation rows) per 4-wave workgroup, 2 x 2 waves of 64 x 64 = 4 x 4 MFMA
// fragments each; v_mfma_f32_16x16x4_f32 (exact f32).  K-step = one 16-column chunk: 8 W fragments + 8 x fragments of
// 1 KiB each.  Because frag16 stores a 16x16 block in the lane order the MFMA consumes, a block goes global -> LDS with ONE
// global_load_lds_dwordx4 per wave (no VGPR round trip, no address arithmetic, no padding) and LDS -> registers with one
// conflict-free ds_read_b128 per lane (lane l reads bytes 16 l .. 16 l + 15 of the block: a linear 1 KiB sweep).
// Pipeline per K-step j (ring of NS = 3 LDS slots, two register sets, ONE barrier per step):
//     wait own LDS-DMA of step j+1 + own LDS reads of step j  ->  barrier  ->  { 64 MFMAs of step j, between them: the 4 LDS-DMA
//     pieces of step j+3 into the slot the barrier released, and the 8 fragment reads of step j+1 into the other register set }
// dealt out 4 MFMAs : 1 memory instruction (sched_group_barrier), so the matrix pipe restarts right behind the barrier and no
// memory instruction is issued outside the shadow of running MFMAs; the DMA has two whole steps to land.
//
// Tiles of all members are one flat list (member, n-tile, m-tile: m fastest), dealt to the XCDs in contiguous runs so the
// workgroups that share an XCD's L2 share W and x panels.  tiles % CUs != 0 would leave the last round partly empty:
// the remainder is cut into `split` k-slabs whose raw accumulators go to a workspace and are finished by k_cond_gemm_fixup
// (slabs added in a fixed order: reproducible).
#pragma once
#include "nd_common.hpp"

#define CG_T 128                 // tile edge (rows of W and rows of x)
#define CG_F 8                   // 16-row fragments per tile edge
#define CG_MAX_SLABS 512         // workspace bound: (remainder tiles) x split <= 512 slabs of 64 KiB

struct CondGemmPlan {
    int use_tile;                // 0: k_skinny handles this shape
    int TM, TN;                  // 128-row tiles over M and N
    int tiles, n_full, rem, split;
    int ntl;                     // MODE 1: partial sums per (row, class): one per 64 output columns
    size_t ws_bytes;
};

// Shapes this kernel takes: more than 128 rows (below that the weight-streaming kernel's one or two passes are faster: a
// 128-row tile would be half empty), fp32 operands, K a multiple of 16.
static inline bool nd_cond_gemm_wanted(int M, int half) { return M > 128 && !half; }

static inline CondGemmPlan nd_cond_gemm_plan(int K, int N, int M, int nm, int half) {
    CondGemmPlan p{};
    p.use_tile = nd_cond_gemm_wanted(M, half) ? 1 : 0;
    const int nfr = (N + 15) / 16, mfr = (M + 15) / 16, nch = K / 16;
    p.TM = (mfr + CG_F - 1) / CG_F;
    p.TN = (nfr + CG_F - 1) / CG_F;
    p.ntl = 2 * p.TN;
    p.tiles = nm * p.TM * p.TN;
    if (!p.use_tile) {          // the weight-streaming kernel takes this shape: no tile rounds, no k-split tail, no workspace
        p.n_full = p.tiles; p.rem = 0; p.split = 1; p.ws_bytes = 0;
        return p;
    }
    const int ncu = nd_num_cus();
    p.n_full = (p.tiles / ncu) * ncu;
    p.rem = p.tiles - p.n_full;
    p.split = 1;
    if (p.rem > 0) {
        // time of the last round in microseconds: a whole tile is nch steps of 64 MFMAs x 32 cycles at ~2 GHz; a split costs
        // the fixup launch (~3 us) plus writing and re-reading the slabs (64 KiB each, ~4 TB/s)
        const double t_tile = nch * 64.0 * 32.0 / 2000.0;
        double best = 1e30;
        const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
        for (int s : cand) {
            if (s > 1 && (nch / s < 16 || (long)p.rem * s > CG_MAX_SLABS)) continue;
            const double rounds = (double)(((long)p.rem * s + ncu - 1) / ncu) / s;
            const double t = rounds * t_tile + (s > 1 ? 3.0 + (double)p.rem * s * 65536.0 * 2.0 / 4.0e6 : 0.0);
            if (t < best - 1e-9) { best = t; p.split = s; }
        }
    }
    if (p.split == 1) { p.n_full = p.tiles; p.rem = 0; }
    p.ws_bytes = p.rem > 0 ? (size_t)p.rem * p.split * CG_T * CG_T * sizeof(float) : 0;
    return p;
}

// The kernels live in nd_cond_gemm.hip, a translation unit of its own: it is built with -mllvm -amdgpu-mfma-vgpr-form so that
// the 64 accumulator registers of a wave stay in the VGPR file (with AGPR accumulators hipcc re-sorts all 64 of them with
// v_accvgpr moves once per loop trip), without touching the code generation of the other kernels.
void* nd_cond_gemm_kernel(int mode);          // kernel handles (MODE 0 / 1 as k_skinny) for hipGraph kernel nodes; argument list:
void* nd_cond_gemm_fixup_kernel(int mode);
size_t nd_cond_gemm_dynlds();                 // dynamic LDS bytes of the launch (0: the staging ring is static)    //   (SkinnyDesc d0, const SkinnyDesc* table, int M, int t, int TM, int TN, int n_full, int split, float* ws)
// d0 / table as nd_launch_skinny.  ws: >= plan.ws_bytes (may be null when plan.rem == 0).
hipError_t nd_launch_cond_gemm(int mode, const CondGemmPlan& p, SkinnyDesc d0, const SkinnyDesc* table, int M, int t, float* ws,
                               hipStream_t st);

// The same blocks on the bf16 matrix pipe with exact fp32 products (frag32b3 operands, csrc/nd_b9.hpp): same argument list, same
// plan (tile list, n_full / split, workspace); launch with 512 threads and nd_cond_gemm_b9_dynlds() bytes of dynamic LDS after one
// nd_cond_gemm_b9_prepare() per device; fixup: rem * 16 workgroups of 64 (one wave per tile, wave sub-tile and row fragment).
void* nd_cond_gemm_b9_kernel(int mode);
void* nd_cond_gemm_b9_fixup_kernel(int mode);
size_t nd_cond_gemm_b9_dynlds();
hipError_t nd_cond_gemm_b9_prepare();
