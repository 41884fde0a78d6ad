// nd_cond_gemm.hip -- the LDS-tiled large-M ConditionalLinear kernels (design notes: nd_cond_gemm.hpp).  gfx950 only.
// Built with -mllvm -amdgpu-mfma-vgpr-form (nested_diffusion_amd/build.py).
#include "nd_cond_gemm.hpp"
#include "nd_b9.hpp"

// which 128 x 128 tile a workgroup owns, and (slab >= 0) which k-slab of it
struct CondGemmTile { int member, tm, tn, slab, rem_index; };

__device__ __forceinline__ CondGemmTile cg_decode(int bid, int n_full, int split, int TM, int TN) {
    CondGemmTile tl;
    tl.slab = -1; tl.rem_index = 0;
    if (bid < n_full) {
        // blocks b and b + 8 share an XCD (round-robin dispatch): XCD x takes the contiguous run of tiles [x*q, (x+1)*q)
        const int q = n_full / 8, r = n_full % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    } else {
        const int j = bid - n_full;
        tl.rem_index = j / split;
        tl.slab = j % split;
        bid = n_full + tl.rem_index;
    }
    const int per = TM * TN;
    tl.member = bid / per;
    const int r2 = bid - tl.member * per;
    tl.tn = r2 / TM;
    tl.tm = r2 - tl.tn * TM;
    return tl;
}

// Epilogue of one wave's 64 x 64 sub-tile.  acc[i][j] = D of (n-fragment i, m-fragment j): lane l holds the 4 consecutive
// output columns n = 16*nf + 4*(l>>4) + r of activation row m = 16*mf + (l&15).
//   MODE 0: v = act(scale[t,n] * acc + shift[t,n]) -> out.  out_packed 1: frag16 [M][N] -- element (m, n) of block (mf, nf)
//           lives at lane (m%16) + 16*((n%16)/4), slot n%4: exactly this lane's float4, one coalesced 1 KiB store per
//           fragment.  out_packed 0: row-major.
//   MODE 1: the same v is not stored; part[m, c, 2*tn + wn] = sum over the wave's 64 columns of pw[c, n] * v  (lin3 +
//           unetnorm3 + softplus + lin4, latent_model.py:181-184); summed r -> i in-lane, then across the 4 lane groups.
template <int MODE, int FI, int FJ>
__device__ __forceinline__ void cg_epilogue_g(f32x4 (&acc)[FI][FJ], const SkinnyDesc& d, int M, int t, int nf0, int mf0, int pcol, int lane,
                                              int ntl) {
    // acc[i][j] = D of (n-fragment nf0 + i, m-fragment mf0 + j); FI fragments = the 64 output columns of one eps partial (pcol)
    const int N = d.N, nfr = (N + 15) >> 4, mfr = (M + 15) >> 4;
    const int g = lane >> 4, li = lane & 15;
    float sc[FI][4], sh[FI][4];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = min((nf0 + i) * 16 + 4 * g + r, N - 1);
            sc[i][r] = d.scale ? nd_ldg(d.scale + (size_t)t * N + n) : 1.0f;
            sh[i][r] = d.shift ? nd_ldg(d.shift + (size_t)t * N + n) : 0.0f;
        }
    const int act = d.act;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = sc[i][r] * acc[i][j][r] + sh[i][r];
                acc[i][j][r] = act == ND_ACT_SOFTPLUS ? nd_softplus(u) : nd_act(u, act);
            }
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int nf = nf0 + i;
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int mf = mf0 + j;
                if (nf < nfr && mf < mfr) {
                    const int n0 = nf * 16 + 4 * g, m = mf * 16 + li;
                    if (d.out_packed == 3) {
                        // frag32b3 image of [M][N] (csrc/nd_b9.hpp): the operand form of the next block on the bf16 matrix pipe
                        if (m < M) nd_b9_store4(reinterpret_cast<bf16x8*>(d.out), N >> 5, m, n0, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    } else if (d.out_packed) {
                        f32x4 v = acc[i][j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n0 + r >= N) v[r] = 0.f;
                        *(__attribute__((address_space(1))) f32x4*)(d.out + ((size_t)mf * nfr + nf) * 256 + lane * 4) = v;
                    } else if (m < M) {
                        float* p = d.out + (size_t)m * N + n0;
                        if (n0 + 3 < N && (N & 3) == 0) *(__attribute__((address_space(1))) f32x4*)p = acc[i][j];
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (n0 + r < N) p[r] = acc[i][j][r];
                        }
                    }
                }
            }
        }
    } else {
        const int C = d.C;
        for (int c = 0; c < C; ++c) {
            float pw[FI][4];
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = (nf0 + i) * 16 + 4 * g + r;
                    pw[i][r] = n < N ? nd_ldg(d.pw + (size_t)c * N + n) : 0.0f;
                }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s += pw[i][r] * acc[i][j][r];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                const int m = (mf0 + j) * 16 + li;
                if (g == 0 && m < M) *(__attribute__((address_space(1))) float*)(d.part + ((size_t)m * C + c) * ntl + pcol) = s;
            }
        }
    }
}

template <int MODE>
__device__ __forceinline__ void cg_epilogue(f32x4 (&acc)[4][4], const SkinnyDesc& d, int M, int t, int tm, int tn, int wn, int wm,
                                            int lane, int ntl) {
    cg_epilogue_g<MODE, 4, 4>(acc, d, M, t, tn * CG_F + wn * 4, tm * CG_F + wm * 4, tn * 2 + wn, lane, ntl);
}

__device__ __forceinline__ const float* cg_uniform_ptr(const float* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}

// NS = LDS slots of the staging ring: the LDS-DMA of step j + NS is issued during step j, so a step's operands have NS - 1
// whole steps to land (NS = 2: 32 KiB of LDS per workgroup, NS = 3: 48 KiB).
template <int MODE, int NS>
__global__ __launch_bounds__(256) void k_cond_gemm(SkinnyDesc d0, const SkinnyDesc* __restrict__ table, int M, int t, int TM, int TN,
                                                   int n_full, int split, float* __restrict__ ws) {
    // ONE LDS object (a second one beside an LDS-DMA target can make hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) float lds[NS][2 * CG_F][256];     // NS slots x (8 W + 8 x fragments) x 1 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;            // this wave's 64 x 64 sub-tile
    const CondGemmTile tl = cg_decode(blockIdx.x, n_full, split, TM, TN);
    const SkinnyDesc d = table ? table[tl.member] : d0;
    const int K = __builtin_amdgcn_readfirstlane(d.K), N = __builtin_amdgcn_readfirstlane(d.N);     // uniform: keeps the loop scalar
    const int nch = K >> 4, nfr = (N + 15) >> 4, mfr = (M + 15) >> 4;
    const int c0 = tl.slab < 0 ? 0 : (int)((long)tl.slab * nch / split);
    const int c1 = tl.slab < 0 ? nch : (int)((long)(tl.slab + 1) * nch / split);
    const int nk = c1 - c0;

    // staging: wave w brings fragments 4w .. 4w+3 of a slot (waves 0,1: the tile's 8 W fragments; waves 2,3: its 8 x fragments);
    // fragment indices past the matrix edge are clamped (their products are never stored)
    // (operand bases made wave-uniform: the DMA address is then an SGPR pair + a constant per-lane offset, no VALU per piece)
    const float* wbase = cg_uniform_ptr(d.w);
    const float* xbase = cg_uniform_ptr(d.x);
    const float* src[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int idx = wave * 4 + f;
        src[f] = idx < CG_F ? wbase + ((size_t)min(tl.tn * CG_F + idx, nfr - 1) * nch + c0) * 256
                            : xbase + ((size_t)min(tl.tm * CG_F + idx - CG_F, mfr - 1) * nch + c0) * 256;
    }
    const int lane4 = lane * 4;
#define CG_STAGE(slot, step)                                                                                              \
    {                                                                                                                     \
        _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                                     \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[f] + (size_t)(step) * 256 + lane4), \
                                             (__attribute__((address_space(3))) void*)&lds[slot][wave * 4 + f][0], 16, 0, 0);       \
    }
#define CG_READ(set, slot)                                                                                                \
    {                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) fw[set][i] = *reinterpret_cast<const f32x4*>(&lds[slot][wn * 4 + i][lane4]);        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) fx[set][j] = *reinterpret_cast<const f32x4*>(&lds[slot][CG_F + wm * 4 + j][lane4]); \
    }
    // one k-quad of a step: 16 independent accumulators (dependent latency 40 > issue interval 32 cycles)
#define CG_MMA_Q(set, q)                                                                                                  \
    {                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                 \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[set][i][q], fx[set][j][q], acc[i][j], 0, 0, 0);       \
    }
    // all but the NS - 2 youngest steps' LDS-DMA of this wave landed (4 per step) + own fragment reads done, then everybody's
#define CG_SYNC()                                                                                                         \
    {                                                                                                                     \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (NS - 2)) : "memory");                                    \
        __builtin_amdgcn_s_barrier();                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fw[2][4], fx[2][4];                          // two register sets, indexed by step parity (static after unrolling)

    // prologue: steps 0 .. NS-1 -> slots 0 .. NS-1 (clamped past the end: valid data nobody reads), step 0 into set 0
#pragma unroll
    for (int u = 0; u < NS; ++u) CG_STAGE(u, min(u, nk - 1))
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    CG_READ(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    // Step j (register set j & 1, slot j % NS): the matrix pipe restarts right behind the barrier; the refill of the slot that
    // barrier released (slot j % NS: its reads were issued in step j-1 and retired in this step's CG_SYNC) with step j + NS, and
    // the fragment reads of step j + 1, are issued between the k-quads, so neither is waited for before 1024+ MFMA cycles.
    constexpr int U = (NS % 2 == 0) ? NS : 2 * NS;     // unroll: slot and register-set indices are compile-time constants
    for (int s = 0; s < nk; s += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s + u < nk) {
                CG_SYNC()
                // one scheduling region: 4 LDS-DMA pieces (slot released by this barrier <- step j + NS), 8 fragment reads
                // (step j + 1 -> the other register set) and the 64 MFMAs of step j; the group barriers below deal them out as
                // 4 MFMAs : 1 memory instruction, MFMAs first, so every memory instruction issues in the shadow of running MFMAs
                CG_STAGE(u % NS, min(s + u + NS, nk - 1))
                CG_READ((u + 1) & 1, (u + 1) % NS)
                CG_MMA_Q(u & 1, 0) CG_MMA_Q(u & 1, 1) CG_MMA_Q(u & 1, 2) CG_MMA_Q(u & 1, 3)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's use of its LDS
#undef CG_STAGE
#undef CG_READ
#undef CG_MMA_Q
#undef CG_SYNC

    if (tl.slab >= 0) {
        float* pt = ws + ((size_t)tl.rem_index * split + tl.slab) * (CG_T * CG_T) + (size_t)wave * 16 * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) *(__attribute__((address_space(1))) f32x4*)(pt + (i * 4 + j) * 256 + lane4) = acc[i][j];
        return;
    }
    cg_epilogue<MODE>(acc, d, M, t, tl.tm, tl.tn, wn, wm, lane, 2 * TN);
}

// Finishes the k-split tiles: accumulators = sum of the slabs in slab order, then the same epilogue.  One WAVE per workgroup
// (a 64 x 64 sub-tile), 4 workgroups per tile: the kernel is a latency chain of slab reads, so it wants many small workgroups.
template <int MODE>
__global__ __launch_bounds__(64) void k_cond_gemm_fixup(SkinnyDesc d0, const SkinnyDesc* __restrict__ table, int M, int t, int TM,
                                                        int TN, int n_full, int split, const float* __restrict__ ws) {
    const int lane = threadIdx.x, wave = blockIdx.x & 3, ri = blockIdx.x >> 2;
    const int wn = wave >> 1, wm = wave & 1;
    const int bid = n_full + ri, per = TM * TN;
    const int member = bid / per, r2 = bid - member * per, tn = r2 / TM, tm = r2 - tn * TM;
    const SkinnyDesc d = table ? table[member] : d0;
    const float* pt = ws + (size_t)ri * split * (CG_T * CG_T) + (size_t)wave * 16 * 256 + lane * 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = *(const __attribute__((address_space(1))) f32x4*)(pt + (i * 4 + j) * 256);
#pragma unroll 2
    for (int k = 1; k < split; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] += *(const __attribute__((address_space(1))) f32x4*)(pt + (size_t)k * (CG_T * CG_T) + (i * 4 + j) * 256);
    cg_epilogue<MODE>(acc, d, M, t, tm, tn, wn, wm, lane, 2 * TN);
}

// ---- the same blocks on the bf16 matrix pipe with exact fp32 products (csrc/nd_b9.hpp) ---------------------------------------------
// Operands as frag32b3 images: d.w = image of the layer's [N][K] weight (made at nd_load_member), d.x = image of the [M][K]
// activations (written by the step head / by MODE 0 of this kernel: out_packed 3).  8 waves per 128 x 128 tile (2 over the columns x 4
// over the rows, 64 x 32 each), one workgroup per CU, 96 KiB LDS ring of two K-steps of 32; tile list, XCD dealing, k-split tail
// and epilogues exactly as k_cond_gemm.  Measured (tools/ubench_bf16x9.hip, K = 5 members x 640 rows): 600-630 us per launch
// against 800-810 us for the f32-input MFMA form.
#define CG9_FA 4
#define CG9_FB 2
#define CG9_WN 2
#define CG9_WM 4
#define CG9_NS 2
#ifndef CG9_FIX_UNROLL
#define CG9_FIX_UNROLL 8       // slabs requested at a time by the fixup (rocprofv3: 17 us per launch with 4, 10 us with 8; the step
                               // in the graph takes 1250-1252 us either way: the chain was not on its critical path)
#endif
#ifndef CG9_NTW
#define CG9_NTW true           // w panels nontemporal (each is read by the TM row tiles of one XCD round, then never again)
#endif
template <int MODE>
__global__ __launch_bounds__(512) void k_cond_gemm_b9(SkinnyDesc d0, const SkinnyDesc* __restrict__ table, int M, int t, int TM, int TN,
                                                      int n_full, int split, float* __restrict__ ws) {
    constexpr int NW = CG9_WN * CG9_WM, NPC = (CG9_WN * CG9_FA + CG9_WM * CG9_FB) * 3, NP = (NPC + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) bf16x8 lds9[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / CG9_WM, wm = wave % CG9_WM;
    const CondGemmTile tl = cg_decode(blockIdx.x, n_full, split, TM, TN);
    const SkinnyDesc d = table ? table[tl.member] : d0;
    const int K = __builtin_amdgcn_readfirstlane(d.K), N = __builtin_amdgcn_readfirstlane(d.N);
    const int nkb = K >> 5, nfr = (N + 15) >> 4, mfr = (M + 15) >> 4;
    const int c0 = tl.slab < 0 ? 0 : (int)((long)tl.slab * nkb / split);
    const int c1 = tl.slab < 0 ? nkb : (int)((long)(tl.slab + 1) * nkb / split);
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(cg_uniform_ptr(d.w));
    const bf16x8* xbase = reinterpret_cast<const bf16x8*>(cg_uniform_ptr(d.x));
    const bf16x8* src[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int e = min(wave * NP + u, NPC - 1), f = e / 3, pl = e % 3;
        const bf16x8* base = f < CG_F ? wbase + ((size_t)min(tl.tn * CG_F + f, nfr - 1) * nkb + c0) * B9_BLOCK_UNITS
                                      : xbase + ((size_t)min(tl.tm * CG_F + f - CG_F, mfr - 1) * nkb + c0) * B9_BLOCK_UNITS;
        src[u] = base + pl * 64 + lane;
    }
    f32x4 acc[CG9_FA][CG9_FB];
#pragma unroll
    for (int i = 0; i < CG9_FA; ++i)
#pragma unroll
        for (int j = 0; j < CG9_FB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    b9_mainloop<CG9_FA, CG9_FB, CG9_WN, CG9_WM, CG9_NS, CG9_NTW>(acc, src, lds9, c1 - c0, wave, wn, wm, lane);
    if (tl.slab >= 0) {     // [remainder tile][slab][wave][fragment][lane]: 8 waves x 8 fragments x 1 KiB = one 128 x 128 fp32 tile
        float* pt = ws + ((size_t)tl.rem_index * split + tl.slab) * (CG_T * CG_T) + (size_t)wave * (CG9_FA * CG9_FB) * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < CG9_FA; ++i)
#pragma unroll
            for (int j = 0; j < CG9_FB; ++j) *(__attribute__((address_space(1))) f32x4*)(pt + (i * CG9_FB + j) * 256) = acc[i][j];
        return;
    }
    cg_epilogue_g<MODE, CG9_FA, CG9_FB>(acc, d, M, t, tl.tn * CG_F + wn * CG9_FA, tl.tm * CG_F + wm * CG9_FB, tl.tn * 2 + wn, lane, 2 * TN);
}

// One wave per (remainder tile, wave sub-tile, row fragment j): the CG9_FA column fragments of a row fragment stay together (MODE 1
// reduces over them), everything else is spread over as many short waves as possible -- the kernel is a chain of slab reads
// (rocprofv3: 17 us per launch with one wave per sub-tile and four slabs requested at a time, 10 us with eight).  Slabs added in
// slab order: reproducible.
template <int MODE>
__global__ __launch_bounds__(64) void k_cond_gemm_b9_fixup(SkinnyDesc d0, const SkinnyDesc* __restrict__ table, int M, int t, int TM,
                                                           int TN, int n_full, int split, const float* __restrict__ ws) {
    constexpr int NW = CG9_WN * CG9_WM;
    const int lane = threadIdx.x, j = blockIdx.x % CG9_FB, wave = (blockIdx.x / CG9_FB) % NW, ri = blockIdx.x / (CG9_FB * NW);
    const int wn = wave / CG9_WM, wm = wave % CG9_WM;
    const int bid = n_full + ri, per = TM * TN;
    const int member = bid / per, r2 = bid - member * per, tn = r2 / TM, tm = r2 - tn * TM;
    const SkinnyDesc d = table ? table[member] : d0;
    const float* pt = ws + (size_t)ri * split * (CG_T * CG_T) + (size_t)wave * (CG9_FA * CG9_FB) * 256 + j * 256 + lane * 4;
    f32x4 acc[CG9_FA][1];
#pragma unroll
    for (int i = 0; i < CG9_FA; ++i) acc[i][0] = *(const __attribute__((address_space(1))) f32x4*)(pt + (i * CG9_FB) * 256);
#pragma unroll CG9_FIX_UNROLL
    for (int k = 1; k < split; ++k) {
        f32x4 v[CG9_FA];
#pragma unroll
        for (int i = 0; i < CG9_FA; ++i) v[i] = *(const __attribute__((address_space(1))) f32x4*)(pt + (size_t)k * (CG_T * CG_T) + (i * CG9_FB) * 256);
#pragma unroll
        for (int i = 0; i < CG9_FA; ++i) acc[i][0] += v[i];
    }
    cg_epilogue_g<MODE, CG9_FA, 1>(acc, d, M, t, tn * CG_F + wn * CG9_FA, tm * CG_F + wm * CG9_FB + j, tn * 2 + wn, lane, 2 * TN);
}

size_t nd_cond_gemm_b9_dynlds() { return (size_t)CG9_NS * (CG9_WN * CG9_FA + CG9_WM * CG9_FB) * 3 * 1024; }
void* nd_cond_gemm_b9_kernel(int mode) { return mode == 1 ? (void*)k_cond_gemm_b9<1> : (void*)k_cond_gemm_b9<0>; }
void* nd_cond_gemm_b9_fixup_kernel(int mode) { return mode == 1 ? (void*)k_cond_gemm_b9_fixup<1> : (void*)k_cond_gemm_b9_fixup<0>; }
// dynamic LDS above 64 KiB has to be allowed per kernel and device before the first launch (graph kernel nodes included)
hipError_t nd_cond_gemm_b9_prepare() {
    hipError_t e = nd_allow_dynamic_lds((const void*)k_cond_gemm_b9<0>, nd_cond_gemm_b9_dynlds());
    if (e != hipSuccess) return e;
    return nd_allow_dynamic_lds((const void*)k_cond_gemm_b9<1>, nd_cond_gemm_b9_dynlds());
}

// Staging depth 3: 48 KiB of LDS and 160 VGPRs per workgroup, three workgroups resident per CU.  Measured at M = 640, K = 5
// members (800 tiles): 811 / 796 us per launch (lin2 / lin3+lin4) = 132 / 135 TFLOP/s; one workgroup per CU with the same
// code 842 us; before the 4 : 1 MFMA : memory interleave three per CU took 940 us (the co-resident waves' LDS-DMA issue stalled
// each other's matrix pipe).  Steady state (tools/bench_cond_gemm.py, 1536 tiles): 230 us per round of 256 tiles against
// 218 us of pure MFMA time = 95 %; the rest of a launch is ~35 us of ramp + the epilogues of the last resident tiles.
#define CG_SLOTS 3
size_t nd_cond_gemm_dynlds() { return 0; }
void* nd_cond_gemm_kernel(int mode) { return mode == 1 ? (void*)k_cond_gemm<1, CG_SLOTS> : (void*)k_cond_gemm<0, CG_SLOTS>; }
void* nd_cond_gemm_fixup_kernel(int mode) { return mode == 1 ? (void*)k_cond_gemm_fixup<1> : (void*)k_cond_gemm_fixup<0>; }

hipError_t nd_launch_cond_gemm(int mode, const CondGemmPlan& p, SkinnyDesc d0, const SkinnyDesc* table, int M, int t, float* ws,
                               hipStream_t st) {
    int TM = p.TM, TN = p.TN, n_full = p.n_full, split = p.split;
    void* args[] = {&d0, &table, &M, &t, &TM, &TN, &n_full, &split, &ws};
    hipError_t e = hipLaunchKernel(nd_cond_gemm_kernel(mode), dim3((unsigned)(p.n_full + p.rem * p.split)), dim3(256), args, nd_cond_gemm_dynlds(), st);
    if (e != hipSuccess || p.rem == 0) return e;
    return hipLaunchKernel(nd_cond_gemm_fixup_kernel(mode), dim3((unsigned)p.rem * 4), dim3(64), args, 0, st);
}
