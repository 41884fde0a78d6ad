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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

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

// HALF-PRECISION OPERANDS ("frag32h", the fp16 mode).  Same idea with 16-row x 32-column blocks of fp16, again 1 KiB:
// block (r/16, k/32) at BYTE offset ((r/16)*(K/32) + k/32)*1024, element (r, k) inside it at half index
// ((r%16) + 16*((k%32)/8))*8 + k%8 -- lane l of a wave holds the 8 halfs k = 8*(l>>4)..+7 of row l&15, the operand
// shape of v_mfma_f32_16x16x32_f16.  A block is 256 float-sized words like a frag16 block, so the streaming kernels
// address both forms identically with nch = K/32 instead of K/16.  Products are exact, accumulation is fp32.
__host__ __device__ __forceinline__ size_t nd_pkh(int r, int k, int nch32) {   // half index
    return ((size_t)(r >> 4) * nch32 + (k >> 5)) * 512 + (size_t)(((r & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7));
}
static inline size_t nd_packed_bytes_dt(int R, int K, int half) {
    return (size_t)((R + 15) / 16) * 16 * (size_t)K * (half ? 2 : 4);
}

// row-major fp32 [R][K] -> frag32h (round to nearest even); rows R .. 16*ceil(R/16) zero-filled.  16 B per thread.
static __global__ __launch_bounds__(256) void k_pack_rows_h(const float* __restrict__ src, _Float16* __restrict__ dst, int R, int K) {
    const int nch = K >> 5;
    const size_t total = (size_t)((R + 15) / 16) * nch * 64;   // 16-byte pieces
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int rt = (int)(blk / nch), c = (int)(blk % nch);
        const int r = rt * 16 + (lane & 15);
        f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < R) {
            const float* p = src + (size_t)r * K + c * 32 + 8 * (lane >> 4);
            const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            h = f16x8{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
        }
        reinterpret_cast<f16x8*>(dst)[i] = h;
    }
}

// frag32h [R][K] -> row-major fp32 (tests / debugging)
static __global__ __launch_bounds__(256) void k_unpack_rows_h(const _Float16* __restrict__ src, float* __restrict__ dst, int R, int K) {
    const int nch = K >> 5;
    const size_t total = (size_t)((R + 15) / 16) * nch * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int rt = (int)(blk / nch), c = (int)(blk % nch);
        const int r = rt * 16 + (lane & 15);
        if (r < R) {
            const f16x8 h = reinterpret_cast<const f16x8*>(src)[i];
            float* p = dst + (size_t)r * K + c * 32 + 8 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = (float)h[j];
        }
    }
}

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

__device__ __forceinline__ f16x8 nd_as_h8(const float4& v) {   // the 16 bytes of a lane's operand, viewed as 8 halfs
    return __builtin_bit_cast(f16x8, v);
}

template <bool NT>
__device__ __forceinline__ float4 nd_ld16(const float* p) {
    if (NT) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    return *reinterpret_cast<const float4*>(p);
}

// One skinny Linear on packed operands, for `nm` members that share the layer shape (K, N):
//   MODE 0: out[m,n] = act(scale[t,n] * sum_k x[m,k] w[n,k] + shift[t,n])
//   MODE 1: that value is not stored; its projection onto C rows is: part[m,c,tile] = sum_{n in tile} pw[c,n]*v
//           -- lin3 + unetnorm3 + softplus + lin4 (latent_model.py:181-184) in one pass.
//   MODE 2: split-K partial sums, no epilogue: part[slab, m, n] = sum_{k in slab} x[m,k] w[n,k]
struct SkinnyDesc {
    const float* x;      // frag16 [M][K]   (frag32h in the fp16 kernels: opaque 1 KiB blocks either way)
    const float* w;      // frag16 [N][K]   (nn.Linear weight, rows padded to 16 with zeros)
    const float* scale;  // [rows, N] or nullptr (=1)
    const float* shift;  // [rows, N] or nullptr (=0)
    float* out;          // MODE 0: out_packed 0 = row-major fp32 [M][N], 1 = frag16 fp32, 2 = frag32h fp16 (N % 32 == 0)
    const float* pw;     // [C, N]            (MODE 1)
    float* part;         // MODE 1: [M, C, ceil(N/16)];  MODE 2: [S, Mpad, Npad]
    int K, N, C, act, out_packed;
};

// WORK DECOMPOSITION.  The nm*ceil(N/16) 16-column output fragments of a launch are numbered
// member-major and dealt contiguously, NF per workgroup (grid.x); grid.y = 16*MT-row groups; grid.z =
// k-slabs (MODE 2 only).  A wave keeps NF weight fragments and MT activation fragments per 16-float
// k-chunk in registers, so an activation fragment is loaded once per NF weight fragments: bytes pulled
// through L1 per weight byte = 1 + MT/NF.  That ratio, not HBM, is what bounds the NF=1 form (measured:
// W-only 53 us, W + x from L2 85 us, MFMA-only 50 us, for 5 x 67 MB at M = 32); with NF chosen so that the
// grid is one workgroup per CU (NF = fragments / 256) the kernel streams at 4.8-4.9 TB/s.
// The WAVES waves of a workgroup split K (interleaved groups of U chunks) and are summed through LDS in
// a fixed order => bitwise reproducible.  A workgroup whose fragment range crosses a member boundary
// takes a single-buffered path that loads the x of its first and of its last member.
// MFMA 16x16x4 f32: A[i=l&15][k=l>>4] <- W rows, B[k=l>>4][j=l&15] <- x rows, D[i=4*(l>>4)+r][j=l&15];
// lane l's float4 holds k = 4*(l>>4)..+3 of a chunk and element jj feeds MFMA jj (same k permutation on
// both operands).
// H = 1: operands are frag32h (fp16), one v_mfma_f32_16x16x32_f16 per fragment pair and 32-column chunk.
template <int MT, int NF, int WAVES, int U, int MODE, bool NT, int H = 0>
__global__ __launch_bounds__(WAVES * 64) void k_skinny(SkinnyDesc d0, const SkinnyDesc* __restrict__ table, int nm,
                                                       int M, int t, int cps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = table ? table[0].K : d0.K, N = table ? table[0].N : d0.N;
    const int nch = H ? K >> 5 : K >> 4, nfr = (N + 15) >> 4, total = nm * nfr, mtiles = (M + 15) >> 4;
    const int f0 = blockIdx.x * NF;
    const int c0 = MODE == 2 ? blockIdx.z * cps : 0;
    const int c1 = MODE == 2 ? min(c0 + cps, nch) : nch;
    int gidx[NF];
    const float* wp[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int fr = min(f0 + f, total - 1);
        gidx[f] = fr / nfr;
        const float* wb = table ? table[gidx[f]].w : d0.w;
        wp[f] = wb + ((size_t)(fr - gidx[f] * nfr) * nch + c0) * 256 + lane * 4;
    }
    const int gA = gidx[0], gB = gidx[NF - 1];
    const float *xA[MT], *xB[MT];
    {
        const float* xa = table ? table[gA].x : d0.x;
        const float* xb = table ? table[gB].x : d0.x;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const size_t off = ((size_t)min((int)blockIdx.y * MT + mt, mtiles - 1) * nch + c0) * 256 + lane * 4;
            xA[mt] = xa + off;
            xB[mt] = xb + off;
        }
    }
    f32x4 acc[NF][MT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Epilogue operands are fetched NOW (their latency hides under the weight stream): each thread's
    // scale/shift entries for the output elements it will finish, and the lin4 rows of MODE 1 into LDS.
    constexpr int EPT = (MT * 256 + WAVES * 64 - 1) / (WAVES * 64);   // epilogue elements per thread
    float esc[NF][EPT], esh[NF][EPT];
    int eact = 0, eC = 0, epacked = 0;
    __shared__ float pws[MODE == 1 ? NF : 1][MODE == 1 ? 8 : 1][16];
    if (MODE != 2) {
        eact = table ? table[0].act : d0.act;
        eC = table ? table[0].C : d0.C;
        epacked = table ? table[0].out_packed : d0.out_packed;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int fr = min(f0 + f, total - 1), g = gidx[f], n0 = (fr - g * nfr) * 16;
            const float* scp = table ? table[g].scale : d0.scale;
            const float* shp = table ? table[g].shift : d0.shift;
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int e = tid + q * WAVES * 64, l = e & 63, r = (e >> 6) & 3;
                const int n = min(n0 + 4 * (l >> 4) + r, N - 1);
                esc[f][q] = scp ? scp[(size_t)t * N + n] : 1.0f;
                esh[f][q] = shp ? shp[(size_t)t * N + n] : 0.0f;
            }
            if (MODE == 1) {
                const float* pwp = table ? table[g].pw : d0.pw;
                if (tid < eC * 16) {
                    const int c = tid >> 4, nl = tid & 15;
                    pws[f][c][nl] = (n0 + nl < N) ? pwp[(size_t)c * N + n0 + nl] : 0.f;
                }
            }
        }
    }

    const int nck = max(c1 - c0, 0);
    const int ngroups = nck / U;     // full groups of U chunks
    const int ngw = ngroups > wave ? (ngroups - wave + WAVES - 1) / WAVES : 0;
    const int glast = ngroups > 0 ? ngroups - 1 : 0;

    float4 wc[U][NF], xc[U][MT];
    auto LDW = [&](float4 (&w)[U][NF], int grp) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int f = 0; f < NF; ++f) w[u][f] = nd_ld16<NT>(wp[f] + ((size_t)grp * U + u) * 256);
    };
    auto LDX = [&](float4 (&x)[U][MT], const float* const (&xb)[MT], int grp) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) x[u][mt] = nd_ld16<false>(xb[mt] + ((size_t)grp * U + u) * 256);
    };
    if (gA == gB) {
        float4 wn[U][NF], xn[U][MT];
        if (ngw > 0) { LDW(wc, min(wave, glast)); LDX(xc, xA, min(wave, glast)); }
        for (int i = 0; i < ngw; ++i) {
            // prefetch the next group (clamped: the last iteration re-reads a valid group, unused)
            const int gn = min(wave + (i + 1) * WAVES, glast);
            LDW(wn, gn); LDX(xn, xA, gn);
            if (H) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(nd_as_h8(wc[u][f]), nd_as_h8(xc[u][mt]), acc[f][mt], 0, 0, 0);
            } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
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
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int f = 0; f < NF; ++f) wc[u][f] = wn[u][f];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xc[u][mt] = xn[u][mt];
            }
        }
    } else {
        float4 xb[U][MT];
        for (int i = 0; i < ngw; ++i) {
            const int gn = wave + i * WAVES;
            LDW(wc, gn); LDX(xc, xA, gn); LDX(xb, xB, gn);
            if (H) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const bool useB = gidx[f] != gA;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(nd_as_h8(wc[u][f]), nd_as_h8(useB ? xb[u][mt] : xc[u][mt]),
                                                                                acc[f][mt], 0, 0, 0);
                    }
            } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const float wv = j == 0 ? wc[u][f].x : j == 1 ? wc[u][f].y : j == 2 ? wc[u][f].z : wc[u][f].w;
                        const bool useB = gidx[f] != gA;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const float xa = j == 0 ? xc[u][mt].x : j == 1 ? xc[u][mt].y : j == 2 ? xc[u][mt].z : xc[u][mt].w;
                            const float xv = j == 0 ? xb[u][mt].x : j == 1 ? xb[u][mt].y : j == 2 ? xb[u][mt].z : xb[u][mt].w;
                            acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, useB ? xv : xa, acc[f][mt], 0, 0, 0);
                        }
                    }
            }
        }
    }
    // leftover chunks (chunk count not a multiple of U): chunk c goes to wave c % WAVES
    for (int c = ngroups * U + wave; c < nck; c += WAVES) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float4 w4 = *reinterpret_cast<const float4*>(wp[f] + (size_t)c * 256);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
            const bool useB = gidx[f] != gA;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 x4 = *reinterpret_cast<const float4*>((useB ? xB[mt] : xA[mt]) + (size_t)c * 256);
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
                if (H) acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(nd_as_h8(w4), nd_as_h8(x4), acc[f][mt], 0, 0, 0);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[j], acc[f][mt], 0, 0, 0);
                }
            }
        }
    }

    // ---- per fragment: cross-wave reduction (fixed order) + epilogue ----
    __shared__ float red[WAVES][MT][4][64];
    __shared__ __attribute__((aligned(16))) float tile[16 * MT][20];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int fr = f0 + f;
        if (fr < total) {                           // uniform across the workgroup
            if (f > 0) __syncthreads();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][mt][r][lane] = acc[f][mt][r];
            __syncthreads();
            const int g = gidx[f], nfi = fr - g * nfr, n0 = nfi * 16;
            if (MODE == 2) {
                float* partp = table ? table[g].part : d0.part;
                // raw partial sums, [slab][Mp][Np]; lane l of the reduced tile owns n = 4*(l>>4)+r, m = l&15
                const int Mp = mtiles * 16, Np = nfr * 16;
                for (int e = tid; e < MT * 64; e += WAVES * 64) {
                    const int mt = e >> 6, l = e & 63;
                    const int mtg = blockIdx.y * MT + mt;
                    if (mtg < mtiles) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float s = red[0][mt][r][l];
#pragma unroll
                            for (int w = 1; w < WAVES; ++w) s += red[w][mt][r][l];
                            v[r] = s;
                        }
                        *reinterpret_cast<float4*>(partp + ((size_t)blockIdx.z * Mp + mtg * 16 + (l & 15)) * Np + n0 + 4 * (l >> 4)) =
                            make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int e = tid + q * WAVES * 64;
                    if (e < MT * 256) {
                        const int mt = e >> 8, r = (e >> 6) & 3, l = e & 63;
                        float s = red[0][mt][r][l];
#pragma unroll
                        for (int w = 1; w < WAVES; ++w) s += red[w][mt][r][l];
                        const int nl = 4 * (l >> 4) + r, ml = 16 * mt + (l & 15);
                        tile[ml][nl] = (n0 + nl < N) ? nd_act(esc[f][q] * s + esh[f][q], eact) : 0.f;
                    }
                }
                __syncthreads();
                const int m0 = blockIdx.y * 16 * MT;
                if (MODE == 0) {
                    float* outp = table ? table[g].out : d0.out;
                    if (epacked == 2) {
                        // frag32h: this fragment's 16 columns are k-groups (nfi&1)*2 + {0,1} of block (m-tile, nfi/2): lanes
                        // 32*(nfi&1) .. +31 of it, 512 contiguous bytes per m-tile
                        _Float16* outh = reinterpret_cast<_Float16*>(outp);
                        const int nch_o = N >> 5;
                        for (int e = tid; e < MT * 32; e += WAVES * 64) {
                            const int mt = e >> 5, L = e & 31;
                            const int mtg = blockIdx.y * MT + mt;
                            if (mtg < mtiles) {
                                const float* tp = &tile[16 * mt + (L & 15)][8 * (L >> 4)];
                                const f16x8 h = {(_Float16)tp[0], (_Float16)tp[1], (_Float16)tp[2], (_Float16)tp[3],
                                                 (_Float16)tp[4], (_Float16)tp[5], (_Float16)tp[6], (_Float16)tp[7]};
                                *reinterpret_cast<f16x8*>(outh + ((size_t)mtg * nch_o + (nfi >> 1)) * 512 + ((nfi & 1) * 32 + L) * 8) = h;
                            }
                        }
                    } else if (epacked) {
                        // the 16x16 block (m-tile, this fragment) is one contiguous 1 KiB of the frag16 output
                        for (int e = tid; e < MT * 64; e += WAVES * 64) {
                            const int mt = e >> 6, L = e & 63;
                            const int mtg = blockIdx.y * MT + mt;
                            if (mtg < mtiles) {
                                const float4 v = *reinterpret_cast<const float4*>(&tile[16 * mt + (L & 15)][4 * (L >> 4)]);
                                *reinterpret_cast<float4*>(outp + ((size_t)mtg * nfr + nfi) * 256 + L * 4) = v;
                            }
                        }
                    } else {
                        for (int e = tid; e < 16 * MT * 16; e += WAVES * 64) {
                            const int ml = e >> 4, nl = e & 15;
                            const int m = m0 + ml, n = n0 + nl;
                            if (m < M && n < N) outp[(size_t)m * N + n] = tile[ml][nl];
                        }
                    }
                } else {
                    float* partp = table ? table[g].part : d0.part;
                    for (int e = tid; e < 16 * MT * eC; e += WAVES * 64) {
                        const int ml = e / eC, c = e - ml * eC;
                        const int m = m0 + ml;
                        if (m < M) {
                            float s = 0.f;
#pragma unroll
                            for (int nl = 0; nl < 16; ++nl) s += pws[MODE == 1 ? f : 0][MODE == 1 ? c : 0][nl] * tile[ml][nl];
                            partp[((size_t)m * eC + c) * nfr + nfi] = s;
                        }
                    }
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

static __global__ __launch_bounds__(256) void k_splitk_epilogue(SplitKEpiDesc d0, const SplitKEpiDesc* __restrict__ table, int M, int S) {
    const SplitKEpiDesc d = table ? table[blockIdx.z] : d0;
    const int N = d.N, Np = ((N + 15) >> 4) * 16, Mp = ((M + 15) >> 4) * 16;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index over [Mp][Np]
    if (q >= (size_t)Mp * Np / 4) return;
    const int m = (int)(q / (Np / 4)), n = (int)(q % (Np / 4)) * 4;
    const size_t slab = (size_t)Mp * Np;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < S; ++k) {
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
    if (d.out_packed == 2) {
        *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(d.out) + nd_pkh(m, n, N >> 5)) =
            f16x4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
    } else if (d.out_packed) {
        *reinterpret_cast<float4*>(d.out + nd_pk(m, n, N >> 4)) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (m < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < N) d.out[(size_t)m * N + n + r] = o[r];
    }
}

// ---- host helpers ---------------------------------------------------------------------------
static inline int nd_pick_mt(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : 4); }
static inline bool nd_use_splitk(int K) { return K >= 16384; }

struct SkinnyLaunch { void* fn; dim3 grid; dim3 block; int cps; int S; };

// Geometry: NF = fragments per workgroup so that the grid is ~one workgroup per CU (256), at most 5
// (register budget at MT = 2 with 16 waves); 16 waves for MT <= 2, 8 for MT = 4.  Weights larger than what
// the 256 MiB Infinity Cache keeps across consecutive steps are streamed with nontemporal loads.
// MODE 2 (split-K): S k-slabs in grid.z so that (fragment groups) x (row groups) x S >= 1024 workgroups.
template <int MODE>
static inline SkinnyLaunch nd_skinny_launch(int K, int N, int M, int nm, int half = 0) {
    const int mt = nd_pick_mt(M);
    const int nfr = (N + 15) / 16, total = nm * nfr, nch = half ? K / 32 : K / 16;
    const int mgroups = (M + 16 * mt - 1) / (16 * mt);
    const int nfmax = mt == 4 ? 2 : 5;
    int nf = total / 256;
    nf = nf < 1 ? 1 : (nf > nfmax ? nfmax : nf);
    int S = 1, cps = nch;
    if (MODE == 2) {
        nf = total >= 256 ? 4 : (total >= 128 ? 2 : 1);
        if (nf > nfmax) nf = nfmax;
        const int gx = (total + nf - 1) / nf;
        S = (1024 + gx * mgroups - 1) / (gx * mgroups);   // ~4+ workgroups per CU: even finish without a work queue
        if (S < 1) S = 1;
        while (S > 1 && nch / S < 64) --S;          // keep every slab >= 64 chunks (1024 floats) deep
        cps = (nch + S - 1) / S;
        S = (nch + cps - 1) / cps;
    }
    const bool nt = (double)nm * N * (double)K * (half ? 2.0 : 4.0) > 160e6;
    SkinnyLaunch L{nullptr, dim3((total + nf - 1) / nf, mgroups, S), dim3(mt == 4 ? 512 : 1024), cps, S};
#define ND_SK(MTV, NFV, WV, UV)                                                                                     \
    (half ? (nt ? (void*)k_skinny<MTV, NFV, WV, UV, MODE, true, 1> : (void*)k_skinny<MTV, NFV, WV, UV, MODE, false, 1>) \
          : (nt ? (void*)k_skinny<MTV, NFV, WV, UV, MODE, true, 0> : (void*)k_skinny<MTV, NFV, WV, UV, MODE, false, 0>))
    if (mt == 4) {
        L.fn = nf == 1 ? ND_SK(4, 1, 8, 2) : ND_SK(4, 2, 8, 2);
    } else if (mt == 2) {
        switch (nf) {
            case 1: L.fn = ND_SK(2, 1, 16, 2); break;
            case 2: L.fn = ND_SK(2, 2, 16, 2); break;
            case 3: L.fn = ND_SK(2, 3, 16, 2); break;
            case 4: L.fn = ND_SK(2, 4, 16, 1); break;
            default: L.fn = ND_SK(2, 5, 16, 1); break;
        }
    } else {
        switch (nf) {
            case 1: L.fn = ND_SK(1, 1, 16, 2); break;
            case 2: L.fn = ND_SK(1, 2, 16, 2); break;
            case 3: L.fn = ND_SK(1, 3, 16, 2); break;
            case 4: L.fn = ND_SK(1, 4, 16, 2); break;
            default: L.fn = ND_SK(1, 5, 16, 2); break;
        }
    }
#undef ND_SK
    return L;
}

static inline size_t nd_splitk_part_floats(int M, int K, int N, int nm = 1, int half = 0) {
    const SkinnyLaunch L = nd_skinny_launch<2>(K, N, M, nm, half);
    return (size_t)L.S * (size_t)(((M + 15) / 16) * 16) * (size_t)(((N + 15) / 16) * 16);
}

static inline hipError_t nd_launch_skinny(const SkinnyLaunch& L, SkinnyDesc d0, const SkinnyDesc* table, int nm, int M, int t,
                                          hipStream_t st) {
    int cps = L.cps;
    void* args[] = {&d0, &table, &nm, &M, &t, &cps};
    return hipLaunchKernel(L.fn, L.grid, L.block, args, 0, st);
}
