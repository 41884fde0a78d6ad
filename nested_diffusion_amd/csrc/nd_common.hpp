// nd_common.hpp -- shared device code for libnd_hip.so (gfx950 / CDNA4 only).
//
// Skinny-M weight-streaming GEMM building blocks.  Every Linear on the sampling path has a small
// row count M (B*mc images, 32 at the headline config) against 4096..150528-wide weights, so the
// kernels are bound by streaming W once from HBM / Infinity Cache; the f32-input MFMA
// (v_mfma_f32_16x16x4_f32, exact f32) keeps the arithmetic off the VALU and is fast enough to
// follow the stream at M = 32 (16 B/clk/CU of W vs ~10 B/clk/CU of HBM).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ND_ACT_NONE 0
#define ND_ACT_SOFTPLUS 1
#define ND_ACT_RELU 2
#define ND_ACT_GELU 3

__device__ __forceinline__ float nd_softplus(float x) {
    // torch softplus(beta=1, threshold=20): x if x > 20 else log1p(exp(x))
    return x > 20.0f ? x : log1pf(expf(x));
}

__device__ __forceinline__ float nd_act(float v, int act) {
    switch (act) {
        case ND_ACT_SOFTPLUS: return nd_softplus(v);
        case ND_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case ND_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        default: return v;
    }
}

// One fused skinny Linear:  out[m,n] = act(scale[t,n] * sum_k x[m,k] w[n,k] + shift[t,n])
// (MODE 0), or its projection onto C output rows without storing out (MODE 1):
// part[tile,m,c] = sum_{n in tile} pw[c,n] * act(...)   -- lin3 + unetnorm3 + softplus + lin4
// (latent_model.py:181-184) in one pass.
struct SkinnyDesc {
    const float* x;      // [M, K]
    const float* w;      // [N, K]   nn.Linear weight layout
    const float* scale;  // [rows, N] or nullptr (=1)
    const float* shift;  // [rows, N] or nullptr (=0)
    float* out;          // [M, N]            (MODE 0)
    const float* pw;     // [C, N]            (MODE 1)
    float* part;         // [ceil(N/16), M, C] (MODE 1)
    int K, N, C, act;
};

#define ND_SK_U 4  // 16-float k-chunks per software-pipeline stage

// Grid: (ceil(N/16), ceil(M/(16*MT)), members).  One workgroup owns 16 output columns for
// 16*MT rows; its WAVES waves split K (interleaved 64-float groups) and are summed through LDS in
// a fixed order, so results are bitwise reproducible.
// MFMA operand maps (16x16x4 f32): A[i=l&15][k=l>>4] <- W rows, B[k=l>>4][j=l&15] <- x rows,
// D[i=4*(l>>4)+r][j=l&15].  Lane l loads a float4 at k = 16*chunk + 4*(l>>4): element jj of every
// lane feeds MFMA jj, i.e. the k order inside a chunk is permuted identically for A and B.
template <int MT, int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k_skinny_fused(SkinnyDesc d0, const SkinnyDesc* __restrict__ table,
                                                             int M, int t) {
    const SkinnyDesc d = table ? table[blockIdx.z] : d0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * MT;
    const int K = d.K, N = d.N;
    const int kq = 4 * (lane >> 4);
    const int nrow = min(n0 + (lane & 15), N - 1);
    const float* wp = d.w + (size_t)nrow * K + kq;
    const float* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mrow = min(m0 + 16 * mt + (lane & 15), M - 1);
        xp[mt] = d.x + (size_t)mrow * K + kq;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int U = ND_SK_U;
    const int nch = K >> 4;          // 16-float chunks
    const int ngroups = nch / U;     // full groups of U chunks
    const int ngw = ngroups > wave ? (ngroups - wave + WAVES - 1) / WAVES : 0;
    const int glast = ngroups > 0 ? ngroups - 1 : 0;

    float4 wc[U], xc[U][MT], wn[U], xn[U][MT];
    if (ngw > 0) {
        const size_t base = (size_t)min(wave, glast) * (U * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wc[u] = *reinterpret_cast<const float4*>(wp + base + u * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xc[u][mt] = *reinterpret_cast<const float4*>(xp[mt] + base + u * 16);
        }
    }
    for (int i = 0; i < ngw; ++i) {
        // prefetch the next group (clamped: the last iteration re-reads a valid group, unused)
        const size_t base = (size_t)min(wave + (i + 1) * WAVES, glast) * (U * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wn[u] = *reinterpret_cast<const float4*>(wp + base + u * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xn[u][mt] = *reinterpret_cast<const float4*>(xp[mt] + base + u * 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float wv[4] = {wc[u].x, wc[u].y, wc[u].z, wc[u].w};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float xv[4] = {xc[u][mt].x, xc[u][mt].y, xc[u][mt].z, xc[u][mt].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[j], acc[mt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wc[u] = wn[u];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xc[u][mt] = xn[u][mt];
        }
    }
    // leftover chunks (K/16 not a multiple of U): chunk c goes to wave c % WAVES
    for (int c = ngroups * U + wave; c < nch; c += WAVES) {
        const float4 w4 = *reinterpret_cast<const float4*>(wp + (size_t)c * 16);
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float4 x4 = *reinterpret_cast<const float4*>(xp[mt] + (size_t)c * 16);
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[j], acc[mt], 0, 0, 0);
        }
    }

    // ---- cross-wave reduction (fixed order) + fused epilogue ----
    __shared__ float red[WAVES][MT][4][64];
    __shared__ float tile[16 * MT][17];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][mt][r][lane] = acc[mt][r];
    __syncthreads();
    const float* sc = d.scale ? d.scale + (size_t)t * N : nullptr;
    const float* sh = d.shift ? d.shift + (size_t)t * N : nullptr;
    for (int e = tid; e < MT * 256; e += WAVES * 64) {
        const int mt = e >> 8, r = (e >> 6) & 3, l = e & 63;
        float s = red[0][mt][r][l];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += red[w][mt][r][l];
        const int nl = 4 * (l >> 4) + r, ml = 16 * mt + (l & 15);
        const int n = n0 + nl;
        float v = 0.f;
        if (n < N) {
            const float a = sc ? sc[n] : 1.0f;
            const float b = sh ? sh[n] : 0.0f;
            v = nd_act(a * s + b, d.act);
        }
        tile[ml][nl] = v;
    }
    __syncthreads();
    if (MODE == 0) {
        for (int e = tid; e < 16 * MT * 16; e += WAVES * 64) {
            const int ml = e >> 4, nl = e & 15;
            const int m = m0 + ml, n = n0 + nl;
            if (m < M && n < N) d.out[(size_t)m * N + n] = tile[ml][nl];
        }
    } else {
        const int C = d.C;
        for (int e = tid; e < 16 * MT * C; e += WAVES * 64) {
            const int ml = e / C, c = e - ml * C;
            const int m = m0 + ml;
            if (m < M) {
                float s = 0.f;
                const int nmax = min(16, N - n0);
                for (int nl = 0; nl < nmax; ++nl) s += d.pw[(size_t)c * N + n0 + nl] * tile[ml][nl];
                d.part[((size_t)blockIdx.x * M + m) * C + c] = s;
            }
        }
    }
}

// ---- split-K variant for very wide inputs (K = 150528: encoder_x.0 and mapping linear1) ------
struct SplitKDesc {
    const float* x;   // [M, K]
    const float* w;   // [N, K]
    float* part;      // [S, M, N]
    int K, N, S, cps; // cps = 16-float chunks per k-slab
};

#define ND_SPK_NF 2     // 16-row W fragments per wave
#define ND_SPK_WAVES 4  // workgroup n-tile = 16 * NF * WAVES = 128 rows
#define ND_SPK_TILE_N (16 * ND_SPK_NF * ND_SPK_WAVES)

// Grid: 1-D, ntiles * S workgroups (x members in z).  Workgroups with equal blockIdx % 8 share an
// XCD (round-robin dispatch; speed only), so k-slabs are dealt to the 8 XCD groups and every
// n-tile of one slab runs on the same XCD: the x slab is then fetched into that XCD's L2 once.
template <int MT>
__global__ __launch_bounds__(ND_SPK_WAVES * 64) void k_skinny_splitk(SplitKDesc d0, const SplitKDesc* __restrict__ table,
                                                                      int M) {
    const SplitKDesc d = table ? table[blockIdx.z] : d0;
    constexpr int NF = ND_SPK_NF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = d.K, N = d.N, S = d.S;
    const int ntiles = (N + ND_SPK_TILE_N - 1) / ND_SPK_TILE_N;
    // decode (slab, tile): bid = 8*idx + xg ; slab = (idx / ntiles) * 8 + xg ; tile = idx % ntiles
    int slab, tileid;
    if ((S & 7) == 0) {
        const int xg = blockIdx.x & 7, idx = blockIdx.x >> 3;
        slab = (idx / ntiles) * 8 + xg;
        tileid = idx % ntiles;
    } else {
        slab = blockIdx.x / ntiles;
        tileid = blockIdx.x % ntiles;
    }
    const int m0 = blockIdx.y * 16 * MT;
    const int nch = K >> 4;
    const int c0 = slab * d.cps, c1 = min(c0 + d.cps, nch);
    const int kq = 4 * (lane >> 4);
    const int nbase = tileid * ND_SPK_TILE_N + wave * (16 * NF);
    const float* wp[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) wp[f] = d.w + (size_t)min(nbase + 16 * f + (lane & 15), N - 1) * K + kq;
    const float* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = d.x + (size_t)min(m0 + 16 * mt + (lane & 15), M - 1) * K + kq;

    f32x4 acc[NF][MT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int U = 2;
    float4 wc[U][NF], xc[U][MT], wn[U][NF], xn[U][MT];
    const int nsteps = (c1 - c0 + U - 1) / U;   // the last step may re-read a chunk; masked below
    const int clast = c1 - 1;
    if (nsteps > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t off = (size_t)min(c0 + u, clast) * 16;
#pragma unroll
            for (int f = 0; f < NF; ++f) wc[u][f] = *reinterpret_cast<const float4*>(wp[f] + off);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xc[u][mt] = *reinterpret_cast<const float4*>(xp[mt] + off);
        }
    }
    for (int i = 0; i < nsteps; ++i) {
        const int cn = c0 + (i + 1) * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t off = (size_t)min(cn + u, clast) * 16;
#pragma unroll
            for (int f = 0; f < NF; ++f) wn[u][f] = *reinterpret_cast<const float4*>(wp[f] + off);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xn[u][mt] = *reinterpret_cast<const float4*>(xp[mt] + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = (c0 + i * U + u) < c1;   // wave-uniform
            if (live) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const float wv[4] = {wc[u][f].x, wc[u][f].y, wc[u][f].z, wc[u][f].w};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float xv[4] = {xc[u][mt].x, xc[u][mt].y, xc[u][mt].z, xc[u][mt].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[j], acc[f][mt], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int f = 0; f < NF; ++f) wc[u][f] = wn[u][f];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xc[u][mt] = xn[u][mt];
        }
    }
    // D[i = 4*(l>>4)+r (n)][j = l&15 (m)] : each lane owns 4 consecutive n of one row m -> float4 store
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int n = nbase + 16 * f + 4 * (lane >> 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + 16 * mt + (lane & 15);
            if (m < M && n < N) {
                float* p = d.part + ((size_t)slab * M + m) * N + n;
                if (n + 3 < N && (N & 3) == 0) {
                    *reinterpret_cast<float4*>(p) = make_float4(acc[f][mt][0], acc[f][mt][1], acc[f][mt][2], acc[f][mt][3]);
                } else {
                    for (int r = 0; r < 4 && n + r < N; ++r) p[r] = acc[f][mt][r];
                }
            }
        }
    }
}

// out[m,n] = act(scale[n] * sum_s part[s,m,n] + shift[n]); slabs summed in order (reproducible).
struct SplitKEpiDesc {
    const float* part;   // [S, M, N]
    const float* scale;  // [N] or nullptr
    const float* shift;  // [N] or nullptr
    float* out;          // [M, N]
    int N, S, act;
};

static __global__ __launch_bounds__(256) void k_splitk_epilogue(SplitKEpiDesc d0, const SplitKEpiDesc* __restrict__ table, int M) {
    const SplitKEpiDesc d = table ? table[blockIdx.z] : d0;
    const size_t total = (size_t)M * d.N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i % d.N);
    float s = 0.f;
    for (int k = 0; k < d.S; ++k) s += d.part[(size_t)k * total + i];
    const float a = d.scale ? d.scale[n] : 1.0f;
    const float b = d.shift ? d.shift[n] : 0.0f;
    d.out[i] = nd_act(a * s + b, d.act);
}

// ---- host helpers ---------------------------------------------------------------------------
static inline int nd_pick_mt(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : 4); }

// number of k-slabs for the split-K path: enough workgroups for >= 2 per CU, multiple of 8 (XCD
// groups), each slab at least 32 chunks (512 floats) deep.
static inline int nd_pick_splitk(int K, int N) {
    const int nch = K / 16;
    const int ntiles = (N + ND_SPK_TILE_N - 1) / ND_SPK_TILE_N;
    int S = (768 + ntiles - 1) / ntiles;
    S = ((S + 7) / 8) * 8;
    while (S > 8 && nch / S < 32) S -= 8;
    if (nch / S < 1) S = 1;
    return S;
}

static inline bool nd_use_splitk(int K) { return K >= 16384; }
