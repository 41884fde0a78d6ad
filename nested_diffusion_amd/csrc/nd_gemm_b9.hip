// nd_gemm_b9.hip -- fp32 Linear layers of the ViT blocks on the bf16 matrix pipe with exact products (design: nd_b9.hpp):
//     out[m,n] = act(sum_k x[m,k] w[n,k] + bias[n]) + res[m,n]
// (timm 0.4.12 Attention.qkv / proj, Mlp.fc1 / fc2, PatchEmbed.proj as a GEMM over im2col rows; call sites
// classification_train_separately.py:337-346), x and w handed over as frag32b3 images (three exact bf16 pieces per fp32 value).
// gfx950 only; built with -mllvm -amdgpu-mfma-vgpr-form (nested_diffusion_amd/build.py).
//
// Two workgroup shapes over the same main loop, picked per GEMM by nd_b9_plan:
//   * 4 waves, tile 128 (w rows) x 64 (x rows), 72 KiB of LDS: two workgroups per CU, each wave 64 x 32.  Best at K = 768 (qkv, fc1,
//     proj, patch embedding): two independent barriers per CU, and a last round that is half empty runs its lone workgroups at
//     almost twice the speed (they have the SIMDs to themselves), so such tails are left whole.
//   * 8 waves, tile 128 x 128, 96 KiB: one workgroup per CU, the same 64 x 32 per wave; a third fewer bytes staged per MFMA.  Best at
//     K >= 2048 (fc2).
// tiles % slots != 0 leaves the last round partly empty: where that costs more than it saves, the remainder tiles are cut into
// `split` k-slabs, raw accumulators to the caller's workspace, finished by k_b9_fixup (slabs added in order: reproducible).
// Epilogue (kernel and fixup alike): bias, activation, residual, then fp32 row-major store and / or a frag32b3 store of the result
// for the GEMM that consumes it (fc1 -> fc2), so that no separate pass ever splits an activation.
#include "nd_b9.hpp"
#include <cstdlib>
#include "../../include/nested_diffusion.h"

int nd_set_err(int code, const char* fmt, ...);
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return nd_set_err(ND_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#ifndef B9_DIRECT_STORE
#define B9_DIRECT_STORE 0          // 1 (variant builds): fp32 outputs stored straight from the accumulators (rounds 4 - early 5)
#endif

struct B9Epilogue {
    const float* bias;      // [N] or null
    const float* res;       // [M][N] or null (added after the activation)
    float* out;             // [M][N] fp32 row-major or null
    bf16x8* out_split;      // frag32b3 image of [M][N] (N % 32 == 0) or null
    int act;
    bf16x8* att;            // ATT form (the qkv Linear, N = 3 * heads * 64): the per-(image, head) qkv images of nd_b9.hpp, nothing else stored
    B9AttLayout al;
};

// ATT form, q / k columns: the lane's 4 consecutive output columns n0 .. n0+3 (= 4 consecutive d of one head) of token row m
__device__ __forceinline__ void b9_epilogue_att_qk(const B9Epilogue& e, f32x4 a, int m, int n0, int M) {
    if (m >= M) return;
    const int E = e.al.heads * 64;
    const int which = n0 >= E ? 1 : 0, c = n0 - which * E, hd = c >> 6, d = c & 63;
    const int b = m / e.al.ntok, tok = m - b * e.al.ntok;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a[r] + (e.bias ? e.bias[n0 + r] : 0.f);
    bf16x8* blk = e.att + (size_t)(b * e.al.heads + hd) * e.al.units()
                  + (size_t)((which ? e.al.k_block0() : e.al.q_block0()) + (tok >> 4) * 2 + (d >> 5)) * B9_BLOCK_UNITS;
    nd_b9_store4_at(blk, (tok & 15) + 16 * ((d & 31) >> 3), (d & 7) >> 2, v[0], v[1], v[2], v[3]);
}

// ATT form, v columns (accumulators of the SWAP main loop): the lane's 4 consecutive token rows m0 .. m0+3 (ntok % 4 == 0: one image,
// 4 consecutive keys) of output column n -> the permuted V^T block of nd_b9.hpp
__device__ __forceinline__ void b9_epilogue_att_v(const B9Epilogue& e, f32x4 a, int m0, int n, int M) {
    if (m0 >= M) return;
    const int E = e.al.heads * 64;
    const int c = n - 2 * E, hd = c >> 6, d = c & 63;
    const int b = m0 / e.al.ntok, key = m0 - b * e.al.ntok, w = key & 31;
    const float bn = e.bias ? e.bias[n] : 0.f;
    bf16x8* blk = e.att + (size_t)(b * e.al.heads + hd) * e.al.units() + (size_t)(e.al.v_block0() + (d >> 4) * e.al.nkb() + (key >> 5)) * B9_BLOCK_UNITS;
    nd_b9_store4_at(blk, (d & 15) + 16 * ((w & 15) >> 2), w >> 4, a[0] + bn, a[1] + bn, a[2] + bn, a[3] + bn);
}

// bias, activation, residual of the lane's 4 consecutive output columns n0 .. n0+3 of row m (m < M, n0 < N)
__device__ __forceinline__ void b9_epilogue_values(const B9Epilogue& e, f32x4 a, int m, int n0, int N, float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nn = min(n0 + r, N - 1);
        float t = a[r] + (e.bias ? e.bias[nn] : 0.f);
        t = nd_act(t, e.act);
        if (e.res && n0 + r < N) t += e.res[(size_t)m * N + n0 + r];
        v[r] = t;
    }
}

// lane's 4 consecutive output columns n0 .. n0+3 of row m
__device__ __forceinline__ void b9_epilogue(const B9Epilogue& e, f32x4 a, int m, int n0, int M, int N) {
    if (m >= M || n0 >= N) return;
    float v[4];
    b9_epilogue_values(e, a, m, n0, N, v);
    if (e.out) {
        float* p = e.out + (size_t)m * N + n0;
        if (n0 + 3 < N && (N & 3) == 0) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n0 + r < N) p[r] = v[r];
        }
    }
    if (e.out_split) nd_b9_store4(e.out_split, N >> 5, m, n0, v[0], v[1], v[2], v[3]);     // N % 32 == 0 (checked on the host)
}

// split == -1: the workgroups past n_full each take HALF of a remainder tile along its x rows (FB / 2 row fragments per wave, the whole
// K): a last round of half-length workgroups and no partial sums -- the tail form of the two-per-CU shape, whose tiles (K = 768) are
// short against a fixup launch.  split >= 1: k-slabs as described above.
// ATT: the qkv Linear of a ViT block with its output written as the attention's operand images (nd_b9.hpp): the tiles of the v
// columns (n >= 2 * heads * 64; a tile never straddles the boundary: heads * 64 % (WN * FA * 16) == 0, checked on the host) run the main
// loop with the MFMA operand roles exchanged, so that a lane holds 4 consecutive keys of one d -- 8-byte pieces of the V^T image.
template <int FA, int FB, int WN, int WM, int NS, bool ATT = false>
__global__ __launch_bounds__(64 * WN * WM) void k_gemm_b9(const bf16x8* __restrict__ xs, const bf16x8* __restrict__ ws, B9Epilogue ep, int M, int K,
                                                          int N, int n_full, int split, f32x4* __restrict__ part) {
    constexpr int NW = WN * WM, NPC = (WN * FA + WM * FB) * 3, NP = (NPC + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) bf16x8 lds[];     // [NS][WN*FA + WM*FB][3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int TN = (N + WN * FA * 16 - 1) / (WN * FA * 16);
    if constexpr (FB % 2 == 0) {
        if (split < 0 && (int)blockIdx.x >= n_full) {
            constexpr int HB = FB / 2, HPC = (WN * FA + WM * HB) * 3, HP = (HPC + NW - 1) / NW;
            const int j = blockIdx.x - n_full, tile = n_full + (j >> 1), half = j & 1;
            const int tm = tile / TN, tn = tile - tm * TN;
            const int nkb = K >> 5, nfr = (N + 15) >> 4, mfr = (M + 15) >> 4;
            const int mf0 = tm * WM * FB + half * WM * HB;              // first x fragment of this half tile
            const bf16x8* src[HP];
#pragma unroll
            for (int u = 0; u < HP; ++u) {
                const int e = min(wave * HP + u, HPC - 1), f = e / 3, pl = e % 3;
                const bf16x8* base = f < WN * FA ? ws + (size_t)min(tn * WN * FA + f, nfr - 1) * nkb * B9_BLOCK_UNITS
                                                 : xs + (size_t)min(mf0 + f - WN * FA, mfr - 1) * nkb * B9_BLOCK_UNITS;
                src[u] = base + pl * 64 + lane;
            }
            f32x4 acc[FA][HB];
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int jj = 0; jj < HB; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (ATT) {
                if (tn * WN * FA * 16 >= 2 * ep.al.heads * 64) {
                    b9_mainloop<FA, HB, WN, WM, NS, false, true>(acc, src, lds, nkb, wave, wn, wm, lane);
#pragma unroll
                    for (int i = 0; i < FA; ++i)
#pragma unroll
                        for (int jj = 0; jj < HB; ++jj)
                            b9_epilogue_att_v(ep, acc[i][jj], (mf0 + wm * HB + jj) * 16 + 4 * (lane >> 4), ((tn * WN + wn) * FA + i) * 16 + (lane & 15), M);
                } else {
                    b9_mainloop<FA, HB, WN, WM, NS>(acc, src, lds, nkb, wave, wn, wm, lane);
#pragma unroll
                    for (int i = 0; i < FA; ++i)
#pragma unroll
                        for (int jj = 0; jj < HB; ++jj)
                            b9_epilogue_att_qk(ep, acc[i][jj], (mf0 + wm * HB + jj) * 16 + (lane & 15), ((tn * WN + wn) * FA + i) * 16 + 4 * (lane >> 4), M);
                }
                return;
            }
            b9_mainloop<FA, HB, WN, WM, NS>(acc, src, lds, nkb, wave, wn, wm, lane);
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int jj = 0; jj < HB; ++jj)
                    b9_epilogue(ep, acc[i][jj], (mf0 + wm * HB + jj) * 16 + (lane & 15), ((tn * WN + wn) * FA + i) * 16 + 4 * (lane >> 4), M, N);
            return;
        }
    }
    int bid = blockIdx.x, slab = -1, rem_index = 0;
    if (bid < n_full) {
        // blocks b and b + 8 share an XCD (round-robin dispatch): XCD x takes a contiguous run of tiles, n fastest, so the workgroups
        // that share an L2 share an x panel
        const int q = n_full / 8, r = n_full % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    } else {
        const int j = bid - n_full;
        rem_index = j / split; slab = j % split; bid = n_full + rem_index;
    }
    const int tm = bid / TN, tn = bid - tm * TN;
    const int nkb = K >> 5, nfr = (N + 15) >> 4, mfr = (M + 15) >> 4;
    const int c0 = slab < 0 ? 0 : (int)((long)slab * nkb / split), c1 = slab < 0 ? nkb : (int)((long)(slab + 1) * nkb / split);
    const bf16x8* src[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int e = min(wave * NP + u, NPC - 1), f = e / 3, pl = e % 3;
        // fragment indices past the matrix edge are clamped (their products are never stored)
        const bf16x8* base = f < WN * FA ? ws + ((size_t)min(tn * WN * FA + f, nfr - 1) * nkb + c0) * B9_BLOCK_UNITS
                                         : xs + ((size_t)min(tm * WM * FB + f - WN * FA, mfr - 1) * nkb + c0) * B9_BLOCK_UNITS;
        src[u] = base + pl * 64 + lane;
    }
    f32x4 acc[FA][FB];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (ATT) {      // whole tiles only (the host launches no k-slabs in this form)
        if (tn * WN * FA * 16 >= 2 * ep.al.heads * 64) {
            b9_mainloop<FA, FB, WN, WM, NS, false, true>(acc, src, lds, c1 - c0, wave, wn, wm, lane);
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j)
                    b9_epilogue_att_v(ep, acc[i][j], ((tm * WM + wm) * FB + j) * 16 + 4 * (lane >> 4), ((tn * WN + wn) * FA + i) * 16 + (lane & 15), M);
        } else {
            b9_mainloop<FA, FB, WN, WM, NS>(acc, src, lds, c1 - c0, wave, wn, wm, lane);
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j)
                    b9_epilogue_att_qk(ep, acc[i][j], ((tm * WM + wm) * FB + j) * 16 + (lane & 15), ((tn * WN + wn) * FA + i) * 16 + 4 * (lane >> 4), M);
        }
        return;
    }
    b9_mainloop<FA, FB, WN, WM, NS>(acc, src, lds, c1 - c0, wave, wn, wm, lane);
    if (slab >= 0) {     // raw accumulators of a k-slab: [remainder tile][slab][wave][fragment][lane], one coalesced 1 KiB store per fragment
        f32x4* pt = part + (((size_t)rem_index * split + slab) * NW + wave) * (FA * FB) * 64 + lane;
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j) *(__attribute__((address_space(1))) f32x4*)(pt + (i * FB + j) * 64) = acc[i][j];
        return;
    }
    // fp32 row-major output of an interior tile: through LDS, so that the global stores are whole 512-byte row segments.  In the MFMA
    // accumulator a lane holds 4 columns of ONE row and 16 lanes hold 16 different rows: a direct store instruction touches 16 rows x 64 B
    // (half cache lines), which costs 6-12 us per ViT launch against the frag32b3 store's contiguous pieces
    // (tools/bench_gemm_store_pattern.py, profiles/r05_gemm_store_pattern.txt).  The operand ring is free here; row stride 144 floats puts
    // both the fragment-order writes and the row-order reads on every bank exactly four times (no conflicts).
    constexpr int BNc = WN * FA * 16, BMc = WM * FB * 16, SLD = BNc + 16;
    if (ep.out && !ep.out_split && (tm + 1) * BMc <= M && (tn + 1) * BNc <= N && (N & 3) == 0 && (size_t)NS * NPC * 1024 >= (size_t)BMc * SLD * 4 &&
        !B9_DIRECT_STORE) {
        float* stage = reinterpret_cast<float*>(lds);
        // The ring is reused as the staging buffer: every wave's LDS-DMA must have landed and every operand read must have returned
        // before any wave writes here.  b9_mainloop ends on vmcnt(0), but this reuse must not depend on that: wait for both counters
        // explicitly, then a full workgroup barrier with memory ordering (a bare s_barrier gives the compiler neither).
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const int ml = (wm * FB + j) * 16 + (lane & 15), nl = (wn * FA + i) * 16 + 4 * (lane >> 4);
                float v[4];
                b9_epilogue_values(ep, acc[i][j], tm * BMc + ml, tn * BNc + nl, N, v);
                *reinterpret_cast<f32x4*>(stage + ml * SLD + nl) = f32x4{v[0], v[1], v[2], v[3]};
            }
        __syncthreads();
        constexpr int LPR = BNc / 4, RPP = (64 * NW) / LPR;          // lanes per row, rows per pass of the whole workgroup
        const int rr = tid / LPR, cc = (tid - rr * LPR) * 4;
        float* ob = ep.out + (size_t)(tm * BMc) * N + (size_t)tn * BNc + cc;
#pragma unroll
        for (int p0 = 0; p0 < BMc; p0 += RPP) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(stage + (p0 + rr) * SLD + cc);
            *(__attribute__((address_space(1))) f32x4*)(ob + (size_t)(p0 + rr) * N) = q;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j)
            b9_epilogue(ep, acc[i][j], ((tm * WM + wm) * FB + j) * 16 + (lane & 15), ((tn * WN + wn) * FA + i) * 16 + 4 * (lane >> 4), M, N);
}

// Finishes the k-split tiles: one wave per (remainder tile, wave sub-tile, FRAGMENT) -- a chain of `split` dependent 1 KiB reads per
// wave, so the launch wants many short waves (with one wave per 8 fragments it took 18-24 us, more than the k-split saved at
// K = 768) -- slabs added in slab order: reproducible.  4 fragments per 256-thread workgroup.
template <int FA, int FB, int WN, int WM>
__global__ __launch_bounds__(256) void k_b9_fixup(const f32x4* __restrict__ part, B9Epilogue ep, int M, int N, int n_full, int split) {
    constexpr int NW = WN * WM, NF = FA * FB;
    const int lane = threadIdx.x & 63;
    const int fid = blockIdx.x * 4 + (threadIdx.x >> 6);            // (remainder tile, wave, fragment)
    const int f = fid % NF, wave = (fid / NF) % NW, ri = fid / (NF * NW);
    const int i = f / FB, j = f - i * FB;
    const int wn = wave / WM, wm = wave % WM;
    const int TN = (N + WN * FA * 16 - 1) / (WN * FA * 16);
    const int bid = n_full + ri, tm = bid / TN, tn = bid - tm * TN;
    const f32x4* p = part + (((size_t)ri * split * NW + wave) * NF + f) * 64 + lane;
    f32x4 a = *(const __attribute__((address_space(1))) f32x4*)p;
#pragma unroll 8
    for (int k = 1; k < split; ++k) a += *(const __attribute__((address_space(1))) f32x4*)(p + (size_t)k * NW * NF * 64);      // all slabs in flight at once
    b9_epilogue(ep, a, ((tm * WM + wm) * FB + j) * 16 + (lane & 15), ((tn * WN + wn) * FA + i) * 16 + 4 * (lane >> 4), M, N);
}

// x [R][K] fp32 row-major -> frag32b3.  One wave per (16 rows x 32 k) block: a lane reads 32 contiguous bytes of its row and writes
// 16 bytes into each plane (three coalesced 1 KiB stores).  Rows R .. Rpad-1 of the last block are written as zeros.
__global__ __launch_bounds__(256) void k_b9_split_rows(const float* __restrict__ x, bf16x8* __restrict__ out, int R, int K) {
    const int lane = threadIdx.x & 63;
    const long blk = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nkb = K >> 5;
    if (blk >= (long)((R + 15) >> 4) * nkb) return;
    const int rb = (int)(blk / nkb), kb = (int)(blk - (long)rb * nkb);
    const int row = rb * 16 + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
    float v[8];
    if (row < R) {
        const float4 u0 = *reinterpret_cast<const float4*>(x + (size_t)row * K + k0), u1 = *reinterpret_cast<const float4*>(x + (size_t)row * K + k0 + 4);
        v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    bf16x8 h1, h2, h3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 a, b, c;
        nd_b9_split(v[e], a, b, c);
        h1[e] = a; h2[e] = b; h3[e] = c;
    }
    out[(blk * 3 + 0) * 64 + lane] = h1;
    out[(blk * 3 + 1) * 64 + lane] = h2;
    out[(blk * 3 + 2) * 64 + lane] = h3;
}

// frag32b3 -> fp32 row-major (sum of the three pieces; exact): tests and debugging
__global__ __launch_bounds__(256) void k_b9_join_rows(const bf16x8* __restrict__ in, float* __restrict__ x, int R, int K) {
    const int lane = threadIdx.x & 63;
    const long blk = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nkb = K >> 5;
    if (blk >= (long)((R + 15) >> 4) * nkb) return;
    const int rb = (int)(blk / nkb), kb = (int)(blk - (long)rb * nkb);
    const int row = rb * 16 + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
    if (row >= R) return;
    const bf16x8 h1 = in[(blk * 3 + 0) * 64 + lane], h2 = in[(blk * 3 + 1) * 64 + lane], h3 = in[(blk * 3 + 2) * 64 + lane];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[(size_t)row * K + k0 + e] = ((float)h1[e] + (float)h2[e]) + (float)h3[e];
}

// ---- launch plan -----------------------------------------------------------------------------------------------------------------
struct B9Plan { int wide; int tiles, n_full, rem, split, slots; size_t ws_bytes, lds_bytes; };
#define B9_FA 4
#define B9_FB 2
#define B9_NS 2

static B9Plan nd_b9_plan(int M, int K, int N) {
    B9Plan p{};
    p.wide = K >= 2048 ? 1 : 0;                                    // 8 waves, 128 x 128, one workgroup per CU
    const int BM = (p.wide ? 4 : 2) * B9_FB * 16, BN = 2 * B9_FA * 16;
    p.tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int ncu = nd_num_cus();
    p.slots = ncu * (p.wide ? 1 : 2);
    p.lds_bytes = (size_t)B9_NS * (2 * B9_FA + (p.wide ? 4 : 2) * B9_FB) * 3 * 1024;
    p.n_full = p.tiles; p.rem = 0; p.split = 1;
    const int rem = p.tiles % p.slots, nkb = K >> 5;
    if (rem > 0 && p.tiles > p.slots && !p.wide) {
        // two-per-CU shape (K = 768: tiles are short against a fixup launch, k-slabs + fixup cost more than they saved).  A small
        // remainder (at most one half tile per CU) goes out as HALF tiles (split = -1: 2 * rem workgroups of half the x rows, the whole
        // K, final results, no fixup); a larger one is left whole.  Measured on [6272, 768] x N (tools/bench_gemm_split_epilogue.py,
        // half / whole): N = 768 (588 tiles, rem 76) 59.3 / 63.2 us, N = 2304 (rem 228) 155.9 / 151.1, N = 3072 (rem 304) 208.8 / 209.8.
        if (2 * rem <= ncu) { p.split = -1; p.rem = rem; p.n_full = p.tiles - rem; }
    } else if (rem > 0 && p.tiles > p.slots) {
        // wide shape (K >= 2048): cost of the last round in microseconds (1.4 us per K-step and workgroup, ~4 us of prologue + epilogue
        // per workgroup); a k-split adds the fixup launch: ~5 us + the slabs written and read back at ~3 TB/s
        const double t_step = 1.4;
        double best = nkb * t_step + 4.0;
        const int cand[] = {2, 3, 4, 6, 8};
        for (int s : cand) {
            if (nkb / s < 4) continue;
            const long slabs = (long)rem * s;
            const double rounds = (double)((slabs + p.slots - 1) / p.slots);
            const double t = rounds * ((double)nkb / s * t_step + 5.0) + 5.0 + (double)slabs * BM * BN * 4.0 * 2.0 / 3.0e6;
            if (t < best - 1e-9) { best = t; p.split = s; }
        }
        if (p.split > 1) { p.rem = rem; p.n_full = p.tiles - rem; }
    }
    p.ws_bytes = p.split > 1 ? (size_t)p.rem * p.split * BM * BN * sizeof(float) : 0;
    return p;
}

extern "C" size_t nd_split_bytes(int rows, int K) {
    if (rows < 1 || K < 32 || (K % 32)) return 0;
    return nd_b9_bytes(rows, K);
}

extern "C" int nd_split_rows(const float* x_dev, void* out_dev, int rows, int K, void* stream) {
    if (!x_dev || !out_dev) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (rows < 1 || K < 32 || (K % 32)) return nd_set_err(ND_ERR_ARG, "need rows >= 1 and K a positive multiple of 32 (K=%d)", K);
    if (((uintptr_t)x_dev | (uintptr_t)out_dev) & 15) return nd_set_err(ND_ERR_ARG, "tensors must be 16-byte aligned");
    const long nb = (long)((rows + 15) / 16) * (K / 32);
    hipLaunchKernelGGL(k_b9_split_rows, dim3((unsigned)((nb * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, (bf16x8*)out_dev, rows, K);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

extern "C" int nd_join_rows(const void* in_dev, float* x_dev, int rows, int K, void* stream) {
    if (!x_dev || !in_dev) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (rows < 1 || K < 32 || (K % 32)) return nd_set_err(ND_ERR_ARG, "need rows >= 1 and K a positive multiple of 32 (K=%d)", K);
    const long nb = (long)((rows + 15) / 16) * (K / 32);
    hipLaunchKernelGGL(k_b9_join_rows, dim3((unsigned)((nb * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)in_dev, x_dev, rows, K);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

extern "C" size_t nd_gemm_split_workspace_bytes(int M, int K, int N) {
    if (M < 1 || N < 1 || K < 32 || (K % 32)) return 0;
    return nd_b9_plan(M, K, N).ws_bytes;
}

extern "C" int nd_gemm_split(const void* x_split, const void* w_split, const float* bias, const float* res, float* out, void* out_split,
                             int M, int K, int N, int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x_split || !w_split || (!out && !out_split)) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (M < 1 || N < 1 || K < 32 || (K % 32)) return nd_set_err(ND_ERR_ARG, "need M,N >= 1 and K a positive multiple of 32 (K=%d)", K);
    if (out_split && (N % 32)) return nd_set_err(ND_ERR_ARG, "a split (frag32b3) output needs N %% 32 == 0 (N=%d)", N);
    if (act < 0 || act > 3) return nd_set_err(ND_ERR_ARG, "unknown activation %d", act);
    if (((uintptr_t)x_split | (uintptr_t)w_split | (uintptr_t)out | (uintptr_t)out_split | (uintptr_t)res) & 15)
        return nd_set_err(ND_ERR_ARG, "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    B9Plan p = nd_b9_plan(M, K, N);
    if (getenv("ND_B9_WHOLE_TAIL") && p.split < 0) { p.n_full = p.tiles; p.rem = 0; p.split = 1; }     // A/B switch for tools/bench_gemm_split_epilogue.py
    if (p.split > 1 && (!workspace || workspace_bytes < p.ws_bytes || ((uintptr_t)workspace & 15))) {
        // no (or too small / misaligned) workspace: every tile whole -- same results up to summation order, longer tail
        p.n_full = p.tiles; p.rem = 0; p.split = 1;
    }
    B9Epilogue ep{bias, res, out, (bf16x8*)out_split, act, nullptr, B9AttLayout{0, 0}};
    const unsigned grid = (unsigned)(p.n_full + p.rem * (p.split < 0 ? 2 : p.split));
    if (p.wide) {
        auto kern = k_gemm_b9<B9_FA, B9_FB, 2, 4, B9_NS>;
        HIP_CHECK(nd_allow_dynamic_lds((const void*)kern, p.lds_bytes));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), p.lds_bytes, st, (const bf16x8*)x_split, (const bf16x8*)w_split, ep, M, K, N, p.n_full, p.split,
                           (f32x4*)workspace);
        HIP_CHECK(hipGetLastError());
        if (p.rem > 0) hipLaunchKernelGGL((k_b9_fixup<B9_FA, B9_FB, 2, 4>), dim3(p.rem * 8 * B9_FA * B9_FB / 4), dim3(256), 0, st, (const f32x4*)workspace, ep, M, N, p.n_full, p.split);
    } else {
        auto kern = k_gemm_b9<B9_FA, B9_FB, 2, 2, B9_NS>;
        HIP_CHECK(nd_allow_dynamic_lds((const void*)kern, p.lds_bytes));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), p.lds_bytes, st, (const bf16x8*)x_split, (const bf16x8*)w_split, ep, M, K, N, p.n_full, p.split,
                           (f32x4*)workspace);
        HIP_CHECK(hipGetLastError());
        if (p.rem > 0 && p.split > 1) hipLaunchKernelGGL((k_b9_fixup<B9_FA, B9_FB, 2, 2>), dim3(p.rem * 4 * B9_FA * B9_FB / 4), dim3(256), 0, st, (const f32x4*)workspace, ep, M, N, p.n_full, p.split);
    }
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

// ---- the qkv Linear of a ViT block with the attention's operand images as its output -------------------------------------------------
extern "C" size_t nd_qkv_images_bytes(int B, int ntok, int heads) {
    if (B < 1 || ntok < 1 || heads < 1) return 0;
    return (size_t)B * heads * B9AttLayout{ntok, heads}.units() * 16;
}

// shapes the image form supports: whole 4-token runs per image (the V^T stores), q / k / v column ranges on tile boundaries
extern "C" int nd_qkv_images_supported(int ntok, int heads) { return ntok >= 4 && (ntok % 4) == 0 && ntok <= 256 && heads >= 1 && ((heads * 64) % (2 * B9_FA * 16)) == 0; }

extern "C" int nd_gemm_split_qkv(const void* x_split, const void* w_split, const float* bias, void* qkv_images, int B, int ntok, int heads, int K,
                                 void* stream) {
    if (!x_split || !w_split || !qkv_images) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (B < 1 || K < 32 || (K % 32)) return nd_set_err(ND_ERR_ARG, "need B >= 1 and K a positive multiple of 32 (K=%d)", K);
    if (!nd_qkv_images_supported(ntok, heads))
        return nd_set_err(ND_ERR_ARG, "qkv images need ntok %% 4 == 0, ntok <= 256 and heads * 64 a multiple of %d (ntok=%d, heads=%d)", 2 * B9_FA * 16, ntok, heads);
    if (((uintptr_t)x_split | (uintptr_t)w_split | (uintptr_t)qkv_images) & 15) return nd_set_err(ND_ERR_ARG, "tensors must be 16-byte aligned");
    const int M = B * ntok, N = 3 * heads * 64;
    // the two-per-CU shape at every depth, whole tiles or half tiles only (no k-slabs: the fixup kernel has no image form)
    const int BM = 2 * B9_FB * 16, BN = 2 * B9_FA * 16, ncu = nd_num_cus();
    const int tiles = ((M + BM - 1) / BM) * (N / BN), slots = 2 * ncu, rem = tiles % slots;
    int n_full = tiles, nrem = 0, split = 1;
    if (rem > 0 && tiles > slots && 2 * rem <= ncu) { split = -1; nrem = rem; n_full = tiles - rem; }
    const size_t lds_bytes = (size_t)B9_NS * (2 * B9_FA + 2 * B9_FB) * 3 * 1024;
    B9Epilogue ep{bias, nullptr, nullptr, nullptr, ND_ACT_NONE, (bf16x8*)qkv_images, B9AttLayout{ntok, heads}};
    auto kern = k_gemm_b9<B9_FA, B9_FB, 2, 2, B9_NS, true>;
    HIP_CHECK(nd_allow_dynamic_lds((const void*)kern, lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_full + nrem * (split < 0 ? 2 : 1))), dim3(256), lds_bytes, (hipStream_t)stream, (const bf16x8*)x_split,
                       (const bf16x8*)w_split, ep, M, K, N, n_full, split, (f32x4*)nullptr);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}
