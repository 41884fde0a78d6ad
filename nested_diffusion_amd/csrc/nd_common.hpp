// nd_common.hpp -- shared device code for libnd_hip.so (gfx950 / CDNA4 only).
//
// Skinny-M weight-streaming GEMM building blocks.  Every Linear on the sampling path has a small
// row count M (B*mc images, 32 at the headline config) against 4096..150528-wide weights, so the
// kernels are bound by streaming W once from HBM; the f32-input MFMA (v_mfma_f32_16x16x4_f32, exact
// f32) keeps the arithmetic off the VALU and just keeps up with the stream at M = 32.
//
// DATA LAYOUT ("frag16" packing).  Both GEMM operands are K-contiguous matrices A[R][K].  They are
// stored as 16-row x 16-column blocks of 1 KiB, block (r/16, k/16) at float offset
// ((r/16)*(K/16) + k/16)*256, and inside a block element (r, k) at ((r%16) + 16*((k%16)/4))*4 + k%4:
// exactly the order in which the 64 lanes of a wave consume it (lane l = row l&15, k-quad l>>4), so
// every wave-level load is one fully coalesced 1 KiB read and a workgroup streams one contiguous
// region.  Weights are packed once at load time; activations are written packed by the producing
// kernel's epilogue.  Measured on MI355X (tools/ubench_skinny.hip, 5 members x 67 MB): row-major
// operands 2.5 TB/s, packed 3.9 TB/s, packed + nontemporal W loads 4.2 TB/s (plain read: 5.4-6.0).
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

// float offset of element (r, k) of a frag16-packed matrix with nch = K/16 chunks per row
__host__ __device__ __forceinline__ size_t nd_pk(int r, int k, int nch) {
    return ((size_t)(r >> 4) * nch + (k >> 4)) * 256 + (size_t)(((r & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3));
}
static inline size_t nd_packed_floats(int R, int K) { return (size_t)((R + 15) / 16) * 16 * (size_t)K; }

// row-major [R][K] -> frag16; rows R .. 16*ceil(R/16) are zero-filled.  One float4 per thread.
static __global__ __launch_bounds__(256) void k_pack_rows(const float* __restrict__ src, float* __restrict__ dst, int R, int K) {
    const int nch = K >> 4;
    const size_t total = (size_t)((R + 15) / 16) * nch * 64;   // float4 count
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int rt = (int)(blk / nch), c = (int)(blk % nch);
        const int r = rt * 16 + (lane & 15);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) v = *reinterpret_cast<const float4*>(src + (size_t)r * K + c * 16 + 4 * (lane >> 4));
        reinterpret_cast<float4*>(dst)[i] = v;
    }
}

// frag16 [R][K] -> row-major (tests / debugging)
static __global__ __launch_bounds__(256) void k_unpack_rows(const float* __restrict__ src, float* __restrict__ dst, int R, int K) {
    const int nch = K >> 4;
    const size_t total = (size_t)((R + 15) / 16) * nch * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int rt = (int)(blk / nch), c = (int)(blk % nch);
        const int r = rt * 16 + (lane & 15);
        if (r < R) *reinterpret_cast<float4*>(dst + (size_t)r * K + c * 16 + 4 * (lane >> 4)) = reinterpret_cast<const float4*>(src)[i];
    }
}

template <bool NT>
__device__ __forceinline__ float4 nd_ld16(const float* p) {
    if (NT) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    return *reinterpret_cast<const float4*>(p);
}

// One fused skinny Linear on packed operands:
//   MODE 0: out[m,n] = act(scale[t,n] * sum_k x[m,k] w[n,k] + shift[t,n])
//   MODE 1: that value is not stored; its projection onto C rows is: part[m,c,tile] = sum_{n in tile} pw[c,n]*v
//           -- lin3 + unetnorm3 + softplus + lin4 (latent_model.py:181-184) in one pass.
struct SkinnyDesc {
    const float* x;      // frag16 [M][K]
    const float* w;      // frag16 [N][K]   (nn.Linear weight, rows padded to 16 with zeros)
    const float* scale;  // [rows, N] or nullptr (=1)
    const float* shift;  // [rows, N] or nullptr (=0)
    float* out;          // MODE 0: frag16 [M][N] if out_packed else row-major [M][N]
    const float* pw;     // [C, N]            (MODE 1)
    float* part;         // [M, C, ceil(N/16)] (MODE 1)
    int K, N, C, act, out_packed;
};

// Grid: (ceil(N/16), ceil(M/(16*MT)), members).  One workgroup owns 16 output columns for 16*MT rows;
// its WAVES waves split K (interleaved groups of U chunks) and are summed through LDS in a fixed
// order, so results are bitwise reproducible.  MFMA 16x16x4 f32: A[i=l&15][k=l>>4] <- W rows,
// B[k=l>>4][j=l&15] <- x rows, D[i=4*(l>>4)+r][j=l&15]; lane l's float4 holds k = 4*(l>>4)..+3 of a
// chunk and element jj feeds MFMA jj (the k order inside a chunk is permuted identically for A and B).
template <int MT, int WAVES, int U, int MODE, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_skinny_fused(SkinnyDesc d0, const SkinnyDesc* __restrict__ table,
                                                             int M, int t) {
    const SkinnyDesc d = table ? table[blockIdx.z] : d0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = d.K, N = d.N;
    const int nch = K >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * MT;
    const int mtiles = (M + 15) >> 4;
    const float* wp = d.w + (size_t)blockIdx.x * nch * 256 + lane * 4;
    const float* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = d.x + (size_t)min(blockIdx.y * MT + mt, mtiles - 1) * nch * 256 + lane * 4;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ngroups = nch / U;     // full groups of U chunks
    const int ngw = ngroups > wave ? (ngroups - wave + WAVES - 1) / WAVES : 0;
    const int glast = ngroups > 0 ? ngroups - 1 : 0;

    float4 wc[U], xc[U][MT], wn[U], xn[U][MT];
    if (ngw > 0) {
        const size_t base = (size_t)min(wave, glast) * (U * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wc[u] = nd_ld16<NT>(wp + base + u * 256);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xc[u][mt] = nd_ld16<false>(xp[mt] + base + u * 256);
        }
    }
    for (int i = 0; i < ngw; ++i) {
        // prefetch the next group (clamped: the last iteration re-reads a valid group, unused)
        const size_t base = (size_t)min(wave + (i + 1) * WAVES, glast) * (U * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wn[u] = nd_ld16<NT>(wp + base + u * 256);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xn[u][mt] = nd_ld16<false>(xp[mt] + base + u * 256);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float wv[4] = {wc[u].x, wc[u].y, wc[u].z, wc[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float xv = j == 0 ? xc[u][mt].x : j == 1 ? xc[u][mt].y : j == 2 ? xc[u][mt].z : xc[u][mt].w;
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv, acc[mt], 0, 0, 0);
                }
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
        const float4 w4 = *reinterpret_cast<const float4*>(wp + (size_t)c * 256);
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float4 x4 = *reinterpret_cast<const float4*>(xp[mt] + (size_t)c * 256);
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[j], acc[mt], 0, 0, 0);
        }
    }

    // ---- cross-wave reduction (fixed order) + fused epilogue ----
    __shared__ float red[WAVES][MT][4][64];
    __shared__ __attribute__((aligned(16))) float tile[16 * MT][20];
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
        if (d.out_packed) {
            // the 16x16 block (m-tile, this n-tile) is one contiguous 1 KiB of the frag16 output
            const int nchN = N >> 4;
            for (int e = tid; e < MT * 64; e += WAVES * 64) {
                const int mt = e >> 6, L = e & 63;
                const int mtg = blockIdx.y * MT + mt;
                if (mtg < mtiles) {
                    const float4 v = *reinterpret_cast<const float4*>(&tile[16 * mt + (L & 15)][4 * (L >> 4)]);
                    *reinterpret_cast<float4*>(d.out + ((size_t)mtg * nchN + blockIdx.x) * 256 + L * 4) = v;
                }
            }
        } else {
            for (int e = tid; e < 16 * MT * 16; e += WAVES * 64) {
                const int ml = e >> 4, nl = e & 15;
                const int m = m0 + ml, n = n0 + nl;
                if (m < M && n < N) d.out[(size_t)m * N + n] = tile[ml][nl];
            }
        }
    } else {
        const int C = d.C, NTl = gridDim.x;
        for (int e = tid; e < 16 * MT * C; e += WAVES * 64) {
            const int ml = e / C, c = e - ml * C;
            const int m = m0 + ml;
            if (m < M) {
                float s = 0.f;
                const int nmax = min(16, N - n0);
                for (int nl = 0; nl < nmax; ++nl) s += d.pw[(size_t)c * N + n0 + nl] * tile[ml][nl];
                d.part[((size_t)m * C + c) * NTl + blockIdx.x] = s;
            }
        }
    }
}

// ---- split-K variant for very wide inputs (K = 150528: encoder_x.0 and mapping linear1) ------
struct SplitKDesc {
    const float* x;   // frag16 [M][K]
    const float* w;   // frag16 [N][K]
    float* part;      // [S, Mpad, Npad] row-major slabs (Mpad, Npad multiples of 16)
    int K, N, S, cps; // cps = 16-float chunks per k-slab
};

#define ND_SPK_NF 2     // 16-row W fragments per wave
#define ND_SPK_WAVES 4  // workgroup n-tile = 16 * NF * WAVES = 128 rows
#define ND_SPK_TILE_N (16 * ND_SPK_NF * ND_SPK_WAVES)

// Grid: x = ntiles * S workgroups, y = m-groups, z = members.  Workgroups with equal blockIdx % 8
// share an XCD (round-robin dispatch; speed only), so k-slabs are dealt to the 8 XCD groups and every
// n-tile of one slab runs on the same XCD: the x slab is then fetched into that XCD's L2 once.
template <int MT, bool NT>
__global__ __launch_bounds__(ND_SPK_WAVES * 64) void k_skinny_splitk(SplitKDesc d0, const SplitKDesc* __restrict__ table,
                                                                      int M) {
    const SplitKDesc d = table ? table[blockIdx.z] : d0;
    constexpr int NF = ND_SPK_NF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = d.K, N = d.N, S = d.S;
    const int ntiles = (N + ND_SPK_TILE_N - 1) / ND_SPK_TILE_N;
    int slab, tileid;
    if ((S & 7) == 0) {
        const int xg = blockIdx.x & 7, idx = blockIdx.x >> 3;
        slab = (idx / ntiles) * 8 + xg;
        tileid = idx % ntiles;
    } else {
        slab = blockIdx.x / ntiles;
        tileid = blockIdx.x % ntiles;
    }
    const int nch = K >> 4;
    const int c0 = slab * d.cps, c1 = min(c0 + d.cps, nch);
    const int nfr_total = (N + 15) >> 4, mtiles = (M + 15) >> 4;
    const int nf0 = tileid * (NF * ND_SPK_WAVES) + wave * NF;     // first 16-row W fragment of this wave
    const float* wp[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) wp[f] = d.w + (size_t)min(nf0 + f, nfr_total - 1) * nch * 256 + lane * 4;
    const float* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = d.x + (size_t)min(blockIdx.y * MT + mt, mtiles - 1) * nch * 256 + lane * 4;

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
            const size_t off = (size_t)min(c0 + u, clast) * 256;
#pragma unroll
            for (int f = 0; f < NF; ++f) wc[u][f] = nd_ld16<NT>(wp[f] + off);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xc[u][mt] = nd_ld16<false>(xp[mt] + off);
        }
    }
    for (int i = 0; i < nsteps; ++i) {
        const int cn = c0 + (i + 1) * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t off = (size_t)min(cn + u, clast) * 256;
#pragma unroll
            for (int f = 0; f < NF; ++f) wn[u][f] = nd_ld16<NT>(wp[f] + off);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xn[u][mt] = nd_ld16<false>(xp[mt] + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = (c0 + i * U + u) < c1;   // wave-uniform
            if (live) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const float wv = j == 0 ? wc[u][f].x : j == 1 ? wc[u][f].y : j == 2 ? wc[u][f].z : wc[u][f].w;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const float xv = j == 0 ? xc[u][mt].x : j == 1 ? xc[u][mt].y : j == 2 ? xc[u][mt].z : xc[u][mt].w;
                            acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[f][mt], 0, 0, 0);
                        }
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
    // D[i = 4*(l>>4)+r (n)][j = l&15 (m)]: each lane owns 4 consecutive n of one row m -> float4 store
    const int Mp = mtiles * 16, Np = nfr_total * 16;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (nf0 + f < nfr_total) {
            const int n = (nf0 + f) * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int mtg = blockIdx.y * MT + mt;
                if (mtg < mtiles) {
                    const int m = mtg * 16 + (lane & 15);
                    *reinterpret_cast<float4*>(d.part + ((size_t)slab * Mp + m) * Np + n) =
                        make_float4(acc[f][mt][0], acc[f][mt][1], acc[f][mt][2], acc[f][mt][3]);
                }
            }
        }
    }
}

// out[m,n] = act(scale[n] * sum_s part[s,m,n] + shift[n]); slabs summed in order (reproducible).
// One thread per 4 consecutive n; output frag16 (feeds the next skinny GEMM) or row-major.
struct SplitKEpiDesc {
    const float* part;   // [S, Mpad, Npad]
    const float* scale;  // [N] or nullptr
    const float* shift;  // [N] or nullptr
    float* out;          // frag16 [M][N] (N % 16 == 0) or row-major [M][N]
    int N, S, act, out_packed;
};

static __global__ __launch_bounds__(256) void k_splitk_epilogue(SplitKEpiDesc d0, const SplitKEpiDesc* __restrict__ table, int M) {
    const SplitKEpiDesc d = table ? table[blockIdx.z] : d0;
    const int N = d.N, Np = ((N + 15) >> 4) * 16, Mp = ((M + 15) >> 4) * 16;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index over [Mp][Np]
    if (q >= (size_t)Mp * Np / 4) return;
    const int m = (int)(q / (Np / 4)), n = (int)(q % (Np / 4)) * 4;
    const size_t slab = (size_t)Mp * Np;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < d.S; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(d.part + (size_t)k * slab + (size_t)m * Np + n);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nn = min(n + r, N - 1);
        const float a = d.scale ? d.scale[nn] : 1.0f;
        const float b = d.shift ? d.shift[nn] : 0.0f;
        o[r] = (n + r < N) ? nd_act(a * o[r] + b, d.act) : 0.f;
    }
    if (d.out_packed) {
        *reinterpret_cast<float4*>(d.out + nd_pk(m, n, N >> 4)) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (m < M) {
        for (int r = 0; r < 4 && n + r < N; ++r) d.out[(size_t)m * N + n + r] = o[r];
    }
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
static inline size_t nd_splitk_part_floats(int M, int K, int N) {
    return (size_t)nd_pick_splitk(K, N) * (size_t)(((M + 15) / 16) * 16) * (size_t)(((N + 15) / 16) * 16);
}

// Launch geometry for the fused kernel.  Few workgroups (one member, <= 2 per CU): 16 waves split K so
// a CU still has enough loads in flight; many workgroups: 8 waves.  Weights bigger than what the
// 256 MiB Infinity Cache can hold across consecutive steps are streamed with nontemporal loads.
struct SkinnyLaunch { void* fn; dim3 grid; dim3 block; };
template <int MODE>
static inline SkinnyLaunch nd_skinny_launch(int K, int N, int M, int nm) {
    const int mt = nd_pick_mt(M);
    const dim3 grid((N + 15) / 16, (M + 16 * mt - 1) / (16 * mt), nm);
    const long wgs = (long)grid.x * grid.y * grid.z;
    const bool nt = (double)nm * N * (double)K * 4.0 > 160e6;
    const bool wide = wgs <= 512 && mt <= 2;      // MT=4 x 16 waves would need > 64 KB of static LDS
    SkinnyLaunch L{nullptr, grid, dim3(wide ? 1024 : 512)};
#define ND_SK_PICK(MTV)                                                                                         \
    L.fn = wide ? (nt ? (void*)k_skinny_fused<MTV, 16, 2, MODE, true> : (void*)k_skinny_fused<MTV, 16, 2, MODE, false>) \
                : (nt ? (void*)k_skinny_fused<MTV, 8, 4, MODE, true> : (void*)k_skinny_fused<MTV, 8, 4, MODE, false>);
    if (mt == 1) { ND_SK_PICK(1) } else if (mt == 2) { ND_SK_PICK(2) } else {
        L.fn = nt ? (void*)k_skinny_fused<4, 8, 4, MODE, true> : (void*)k_skinny_fused<4, 8, 4, MODE, false>;
    }
#undef ND_SK_PICK
    return L;
}
