// nd_vit.hip -- kernels for the ViT prefix of the mapping network (timm 0.4.12 vit_base_patch16_224
// semantics; call sites classification_train_separately.py:337-340).  gfx950 only, fp32 with
// f32-input MFMA (exact f32 products).
#include "nd_common.hpp"
#include "nd_b9.hpp"
#include "../../include/nested_diffusion.h"
#include <cstdlib>

int nd_set_err(int code, const char* fmt, ...);
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return nd_set_err(ND_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// fp32 GEMM kernel: nd_gemm_f32.hip (its own translation unit, accumulators in VGPRs)
hipError_t nd_launch_gemm_nt_128x64(const float* x, const float* w, const float* bias, const float* res, float* out, int M, int K, int N,
                                    int act, int n_full, int split, float* part, unsigned grid, hipStream_t st);
// ---------------------------------------------------------------------------------------------
// Large-M GEMM, both operands K-contiguous:  out[m,n] = act(sum_k x[m,k] w[n,k] + bias[n]) + res[m,n]
// The fp32 kernel (k_gemm_nt<128,64>: 3-stage pipeline, one barrier per K-step) lives in nd_gemm_f32.hip; this file holds the
// launch plan, the k-split fixup and the fp16-operand form.
// ---------------------------------------------------------------------------------------------
#define GB_K 16
#define GB_LD 24   // 16 + 8 pad: ds_read_b128 of (row = l&15, k-quad = l>>4) is bank-conflict-free

// ---------------------------------------------------------------------------------------------
// fp16-operand form of the same GEMM (the fp16 mode, BASELINE config 5; not a mode of the reference):
// x stays fp32 in HBM and is rounded to fp16 on its way into LDS, w is an fp16 [N][K] copy made once at load; products are
// exact, accumulation / bias / activation / residual fp32, out fp32.  BK = 32 per stage = one v_mfma_f32_16x16x32_f16 per
// fragment pair; LDS rows are 32 halfs + 8 pad (80 B: the 16 rows of a fragment start at banks 20r mod 64, all distinct,
// so a lane's 8 halfs k = 8*(l>>4)..+7 are one conflict-free ds_read_b128).  Same tile order, tail split and fixup.
// ---------------------------------------------------------------------------------------------
#define GH_K 32
#define GH_LD 40
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm_h(const float* __restrict__ x, const _Float16* __restrict__ w,
                                                const float* __restrict__ bias, const float* __restrict__ res,
                                                float* __restrict__ out, int M, int K, int N, int act, int n_full, int split,
                                                float* __restrict__ part) {
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int LA = BM * GH_K / 4 / 256;      // float4 (4 fp32) loads per thread for the x tile
    constexpr int LB = BN * GH_K / 8 / 256;      // 16-byte (8 halfs) loads per thread for the w tile
    static_assert(LA >= 1 && LB >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) _Float16 sA[2][BM][GH_LD];
    __shared__ __attribute__((aligned(16))) _Float16 sB[2][BN][GH_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    int bid = blockIdx.x, slab = -1;
    if (bid < n_full) {
        const int q = n_full / 8, r = n_full % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    } else {
        const int j = bid - n_full;
        bid = n_full + j / split;
        slab = j % split;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[LA];
    f16x8 rb[LB];
#define GH_GLOAD(k0)                                                                                      \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                  \
            const int e = tid + i * 256, row = e >> 3, kq = (e & 7) * 4;                                  \
            ra[i] = *reinterpret_cast<const float4*>(x + (size_t)min(m0 + row, M - 1) * K + (k0) + kq);  \
        }                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) {                                                  \
            const int e = tid + i * 256, row = e >> 2, kq = (e & 3) * 8;                                  \
            rb[i] = *reinterpret_cast<const f16x8*>(w + (size_t)min(n0 + row, N - 1) * K + (k0) + kq);   \
        }                                                                                                 \
    }
#define GH_SWRITE(buf)                                                                                    \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                  \
            const int e = tid + i * 256, row = e >> 3, kq = (e & 7) * 4;                                  \
            *reinterpret_cast<f16x4*>(&sA[buf][row][kq]) =                                                \
                f16x4{(_Float16)ra[i].x, (_Float16)ra[i].y, (_Float16)ra[i].z, (_Float16)ra[i].w};        \
        }                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) {                                                  \
            const int e = tid + i * 256, row = e >> 2, kq = (e & 3) * 8;                                  \
            *reinterpret_cast<f16x8*>(&sB[buf][row][kq]) = rb[i];                                         \
        }                                                                                                 \
    }
    const int nkt = K / GH_K;
    const int ks0 = slab < 0 ? 0 : (int)((long)slab * nkt / split);
    const int nk = slab < 0 ? nkt : (int)((long)(slab + 1) * nkt / split);
    GH_GLOAD(ks0 * GH_K)
    GH_SWRITE(ks0 & 1)
    __syncthreads();
    const int lr = lane & 15, lk = 8 * (lane >> 4);
    for (int ks = ks0; ks < nk; ++ks) {
        const int buf = ks & 1;
        GH_GLOAD(min(ks + 1, nk - 1) * GH_K)
        f16x8 fa[FM], fb[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const f16x8*>(&sA[buf][wr * WM + 16 * i + lr][lk]);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const f16x8*>(&sB[buf][wc * WN + 16 * j + lr][lk]);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D[i=n][j=m]
        GH_SWRITE(buf ^ 1)
        __syncthreads();
    }
#undef GH_GLOAD
#undef GH_SWRITE
    if (slab >= 0) {
        float* pt = part + ((size_t)(bid - n_full) * split + slab) * (BM * BN);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<float4*>(pt + (wr * WM + 16 * i + (lane & 15)) * BN + wc * WN + 16 * j + 4 * (lane >> 4)) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wr * WM + 16 * i + (lane & 15);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wc * WN + 16 * j + 4 * (lane >> 4);
            if (m < M && n < N) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nn = min(n + r, N - 1);
                    float t = acc[i][j][r] + (bias ? bias[nn] : 0.f);
                    t = nd_act(t, act);
                    if (res && n + r < N) t += res[(size_t)m * N + n + r];
                    v[r] = t;
                }
                float* p = out + (size_t)m * N + n;
                if (n + 3 < N && (N & 3) == 0) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) p[r] = v[r];
                }
            }
        }
    }
}

// Finishes the k-split tiles: out = act(sum_slabs part + bias) + res, slabs added in order (reproducible).
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm_fixup(const float* __restrict__ part, const float* __restrict__ bias,
                                                    const float* __restrict__ res, float* __restrict__ out, int M, int N, int act,
                                                    int n_full, int split) {
    const int tiles_n = (N + BN - 1) / BN;
    const int tile = n_full + blockIdx.y;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int e = blockIdx.x * 256 + threadIdx.x;           // float4 index inside the tile
    const int ml = e / (BN / 4), nl = (e % (BN / 4)) * 4;
    const int m = m0 + ml, n = n0 + nl;
    if (m >= M || n >= N) return;
    const float* pt = part + (size_t)blockIdx.y * split * (BM * BN) + (size_t)ml * BN + nl;
    float4 s4 = *reinterpret_cast<const float4*>(pt);
    for (int k = 1; k < split; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(pt + (size_t)k * (BM * BN));
        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    float v[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nn = min(n + r, N - 1);
        float t = v[r] + (bias ? bias[nn] : 0.f);
        t = nd_act(t, act);
        if (res && n + r < N) t += res[(size_t)m * N + n + r];
        v[r] = t;
    }
    float* p = out + (size_t)m * N + n;
    if (n + 3 < N && (N & 3) == 0) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < N) p[r] = v[r];
    }
}

// Launch plan.  Measured (tools/bench_gemm2.py): the run time of a launch is ceil(tiles / 256) x (time of one tile), i.e. it
// is quantised on whole tiles per CU, and the 128x64 tile is as fast as 128x128 per flop.  So: 128x64 tiles; the
// tiles % 256 remainder is cut into `split` k-slabs where that shortens the last round (needs the workspace).
#define GT_BM 128
#define GT_BN 64
#define GT_CUS (nd_num_cus())
struct GemmPlan { int tiles, n_full, rem, split; size_t ws_bytes; };
static GemmPlan nd_gemm_plan(int M, int K, int N, int bk = GB_K) {
    GemmPlan p{};
    p.tiles = ((M + GT_BM - 1) / GT_BM) * ((N + GT_BN - 1) / GT_BN);
    const int ncu = GT_CUS;
    p.n_full = (p.tiles / ncu) * ncu;
    p.rem = p.tiles - p.n_full;
    p.split = 1;
    const int nk = K / bk;
    if (p.rem > 0 && (size_t)M * N * K >= ((size_t)1 << 28)) {
        // tail length in tile-times: ceil(rem * s / 256) / s ; take the smallest s that gets within 10 % of the best
        double best = 1e9;
        const int cand[] = {1, 2, 3, 4, 5, 6, 8};
        for (int s : cand) if (nk / s >= 8) best = fmin(best, (double)((p.rem * s + ncu - 1) / ncu) / s);
        for (int s : cand)
            if (nk / s >= 8 && (double)((p.rem * s + ncu - 1) / ncu) / s <= best * 1.1 + 1e-9) { p.split = s; break; }
    }
    p.ws_bytes = p.split > 1 ? (size_t)p.rem * p.split * GT_BM * GT_BN * sizeof(float) : 0;
    return p;
}

extern "C" size_t nd_gemm_workspace_bytes(int M, int K, int N, int dtype) {
    const int km = dtype == ND_DTYPE_F16 ? GH_K : GB_K;
    if ((dtype != ND_DTYPE_F32 && dtype != ND_DTYPE_F16) || M < 1 || N < 1 || K < km || (K % km)) return 0;
    return nd_gemm_plan(M, K, N, km).ws_bytes;
}

extern "C" int nd_gemm_bias_act(const float* x, const void* w, const float* bias, const float* res, float* out, int M, int K,
                                int N, int act, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !w || !out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (dtype != ND_DTYPE_F32 && dtype != ND_DTYPE_F16) return nd_set_err(ND_ERR_ARG, "unknown dtype %d", dtype);
    const int half = dtype == ND_DTYPE_F16, km = half ? GH_K : GB_K;
    if (M < 1 || N < 1 || K < km || (K % km)) return nd_set_err(ND_ERR_ARG, "need M,N >= 1 and K a positive multiple of %d (K=%d)", km, K);
    if (act < 0 || act > 3) return nd_set_err(ND_ERR_ARG, "unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    GemmPlan p = nd_gemm_plan(M, K, N, km);
    if (p.split > 1 && (!workspace || workspace_bytes < p.ws_bytes || ((uintptr_t)workspace & 15))) {
        // no (or too small / misaligned) workspace: every tile whole -- same results up to summation order, longer tail
        p.n_full = p.tiles; p.rem = 0; p.split = 1;
    }
    if (p.split == 1) { p.n_full = p.tiles; p.rem = 0; }
    float* part = (float*)workspace;
    const dim3 grid((unsigned)(p.n_full + p.rem * p.split));
    if (half)
        hipLaunchKernelGGL((k_gemm_h<GT_BM, GT_BN>), grid, dim3(256), 0, st, x, (const _Float16*)w, bias, res, out, M, K, N, act, p.n_full,
                           p.split, part);
    else
        HIP_CHECK(nd_launch_gemm_nt_128x64(x, (const float*)w, bias, res, out, M, K, N, act, p.n_full, p.split, part, grid.x, st));
    HIP_CHECK(hipGetLastError());
    if (p.rem > 0) {
        hipLaunchKernelGGL((k_gemm_fixup<GT_BM, GT_BN>), dim3(GT_BM * GT_BN / 4 / 256, p.rem), dim3(256), 0, st, part, bias, res, out, M, N,
                           act, p.n_full, p.split);
        HIP_CHECK(hipGetLastError());
    }
    return ND_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim; one wave per row, two-pass (mean, then centred variance) in registers.
// ---------------------------------------------------------------------------------------------
// SPLIT: the result is written as a frag32b3 image (csrc/nd_b9.hpp) -- the input form of the Linear layer that follows every
// LayerNorm of a ViT block -- instead of fp32 row-major: a lane's 4 consecutive columns are three 8-byte pieces.
template <int VPL, bool SPLIT>  // float4 per lane: dim <= 64*4*VPL
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ out, int rows, int dim,
                                                   float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + (size_t)row * dim;
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < dim ? *reinterpret_cast<const float4*>(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < dim) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < dim) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float4 b = *reinterpret_cast<const float4*>(beta + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (SPLIT) nd_b9_store4(reinterpret_cast<bf16x8*>(out), dim >> 5, row, c, o.x, o.y, o.z, o.w);
            else *reinterpret_cast<float4*>(out + (size_t)row * dim + c) = o;
        }
    }
}

// LayerNorm -> frag32b3 for 16 rows per workgroup (16 waves, one row each): the pieces are gathered in LDS as the block-row's image --
// which is ONE contiguous run of (dim/32) * 3 KiB in global memory -- and copied out in coalesced 16-byte units.  (The direct form
// above scatters 8-byte pieces 256 bytes apart: ~15 us per [6272, 768] launch; this one 13.5 us against 8.1 us for the fp32 LayerNorm
// and a floor of ~10 us for 19 MB read + 29 MB written; with 4 waves x 4 rows per workgroup it took 18 us: a row is two dependent
// shuffle reductions, four of them in a row are a latency chain.  tools/bench_ln_split.py)
template <int VPL>
__global__ __launch_bounds__(1024) void k_layernorm_split16(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16x8* __restrict__ out, int rows, int dim, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ln_img[];      // [dim/32][3][64][16 B]
    const int lane = threadIdx.x & 63, r16 = threadIdx.x >> 6;
    const int nkb = dim >> 5;
    {
        const int row = blockIdx.x * 16 + r16;
        float4 v[VPL];
        float s = 0.f;
        const bool live = row < rows;
        const float* p = x + (size_t)(live ? row : 0) * dim;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            v[i] = (live && c < dim) ? *reinterpret_cast<const float4*>(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < dim) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
        const float rstd = 1.0f / sqrtf(q / (float)dim + eps);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < dim) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + c);
                const float4 b = *reinterpret_cast<const float4*>(beta + c);
                float o[4];
                o[0] = live ? (v[i].x - mean) * rstd * g.x + b.x : 0.f;
                o[1] = live ? (v[i].y - mean) * rstd * g.y + b.y : 0.f;
                o[2] = live ? (v[i].z - mean) * rstd * g.z + b.z : 0.f;
                o[3] = live ? (v[i].w - mean) * rstd * g.w + b.w : 0.f;
                bf16x4 p1, p2, p3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 h1, h2, h3;
                    nd_b9_split(o[e], h1, h2, h3);
                    p1[e] = h1; p2[e] = h2; p3[e] = h3;
                }
                unsigned char* q8 = ln_img + (size_t)(c >> 5) * 3072 + (r16 + 16 * ((c & 31) >> 3)) * 16 + ((c & 7) >> 2) * 8;
                *reinterpret_cast<bf16x4*>(q8) = p1;
                *reinterpret_cast<bf16x4*>(q8 + 1024) = p2;
                *reinterpret_cast<bf16x4*>(q8 + 2048) = p3;
            }
        }
    }
    __syncthreads();
    const uint4* src = reinterpret_cast<const uint4*>(ln_img);
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * nkb * B9_BLOCK_UNITS);
    for (int i = threadIdx.x; i < nkb * B9_BLOCK_UNITS; i += 1024) dst[i] = src[i];
}

template <bool SPLIT>
static int launch_layernorm(const float* x, const float* gamma, const float* beta, float* out, int rows, int dim, float eps, void* stream) {
    if (!x || !gamma || !beta || !out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (rows < 1 || dim < 4 || (dim % 4) || dim > 64 * 4 * 8) return nd_set_err(ND_ERR_ARG, "dim must be a multiple of 4 in [4,2048]");
    if (SPLIT && (dim % 32)) return nd_set_err(ND_ERR_ARG, "a split (frag32b3) output needs dim %% 32 == 0 (dim=%d)", dim);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((rows + 3) / 4), block(256);
    const int vpl = (dim + 255) / 256;
    if (SPLIT && rows >= 64 && dim <= 1024) {
        // 16 rows per workgroup through an LDS image (<= 96 KiB): coalesced stores
        const dim3 g16((rows + 15) / 16);
        const size_t lds = (size_t)(dim / 32) * 3072;
        bf16x8* img = reinterpret_cast<bf16x8*>(out);
#define LN16(V)                                                                                                                      \
        {                                                                                                                            \
            HIP_CHECK(nd_allow_dynamic_lds((const void*)k_layernorm_split16<V>, 96 * 1024));                                         \
            hipLaunchKernelGGL((k_layernorm_split16<V>), g16, dim3(1024), lds, st, x, gamma, beta, img, rows, dim, eps);                  \
        }
        if (vpl <= 1) LN16(1) else if (vpl <= 2) LN16(2) else if (vpl <= 3) LN16(3) else LN16(4)
#undef LN16
        HIP_CHECK(hipGetLastError());
        return ND_OK;
    }
    if (vpl <= 1) hipLaunchKernelGGL((k_layernorm<1, SPLIT>), grid, block, 0, st, x, gamma, beta, out, rows, dim, eps);
    else if (vpl <= 2) hipLaunchKernelGGL((k_layernorm<2, SPLIT>), grid, block, 0, st, x, gamma, beta, out, rows, dim, eps);
    else if (vpl <= 3) hipLaunchKernelGGL((k_layernorm<3, SPLIT>), grid, block, 0, st, x, gamma, beta, out, rows, dim, eps);
    else if (vpl <= 4) hipLaunchKernelGGL((k_layernorm<4, SPLIT>), grid, block, 0, st, x, gamma, beta, out, rows, dim, eps);
    else hipLaunchKernelGGL((k_layernorm<8, SPLIT>), grid, block, 0, st, x, gamma, beta, out, rows, dim, eps);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

extern "C" int nd_layernorm(const float* x, const float* gamma, const float* beta, float* out, int rows, int dim, float eps,
                            void* stream) {
    return launch_layernorm<false>(x, gamma, beta, out, rows, dim, eps, stream);
}

// the same LayerNorm with its result written as the frag32b3 image of [rows, dim] (dim % 32 == 0): the input of nd_gemm_split
extern "C" int nd_layernorm_split(const float* x, const float* gamma, const float* beta, void* out_split, int rows, int dim, float eps,
                                  void* stream) {
    return launch_layernorm<true>(x, gamma, beta, (float*)out_split, rows, dim, eps, stream);
}

// ---------------------------------------------------------------------------------------------
// Attention core for d = 64:  out = softmax(q k^T / 8) v per (image, head), q/k/v read from the fused qkv activations.
// MFMA operand maps shared by the fp32 kernel (nd_attention.hip) and the fp16-operand kernel below:
//   S^T = K Q^T   : A = K[key=16f+(l&15)][d=16c+4g+jj], B = Q[q=l&15][same d]  -> D[key=16f+4g+r][q=l&15]
//   softmax over keys: in-lane over (f, r), across the 4 lane groups g by xor 16/32.
//   O^T = V^T P^T : the k-step (f, r) takes key 16f+4g+r from lane group g -- exactly the S^T register acc[f][r] of that
//                   lane -- and A = V[key][4*(l&15)+e] (float4, e = d-frag), so P needs no transpose.  D_e[i=4g'+r'][q]: d = 4*i + e.
// ---------------------------------------------------------------------------------------------
#define AT_MAXF 16  // up to 256 keys
// RING form (the default fp32 kernel): nd_attention.hip, a translation unit of its own (accumulators kept in VGPRs).
hipError_t nd_launch_attention_ring(const float* qkv, float* out, int B, int N, int heads, int split_out, hipStream_t st);
hipError_t nd_launch_attention_b9(const void* att, float* out, int B, int N, int heads, int split_out, hipStream_t st);      // nd_attention.hip
extern "C" int nd_qkv_images_supported(int ntok, int heads);                                                                   // nd_gemm_b9.hip

// fp16-operand form (the fp16 mode; not a mode of the reference): q, k, v are rounded to fp16 as they are staged, both
// contractions run on v_mfma_f32_16x16x32_f16 with fp32 accumulation, the softmax is fp32 and the normalised probabilities
// are rounded to fp16 for the second contraction.  K is staged row-major [key][64+8] (a lane's 8 halfs d = 8g..+7 of key
// l&15: one ds_read_b128), V TRANSPOSED [d][keys+8] so that the 8 k-slots of a lane are two runs of 4 consecutive keys
// {32F+4g..+3} and {32F+16+4g..+3} -- exactly the keys whose scores that lane already holds in the accumulators of score
// fragments 2F and 2F+1 (D[i=4g+r][j=query]), so P needs no cross-lane movement.
#define AH_KLD 72
template <int NF>
__global__ __launch_bounds__(NF * 64) void k_attention_h(const float* __restrict__ qkv, float* __restrict__ out, int B, int N,
                                                          int heads) {
    constexpr int NFP = (NF + 1) / 2 * 2;                 // score fragments, padded to pairs
    constexpr int VLD = 16 * NFP + 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* sK = reinterpret_cast<_Float16*>(smem);     // [16*NFP][AH_KLD]
    _Float16* sVt = sK + 16 * NFP * AH_KLD;               // [64][VLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = blockIdx.y * (blockDim.x >> 6) + (tid >> 6);
    const int bh = blockIdx.x, b = bh / heads, hd = bh % heads;
    const int Cm = heads * 64;
    const size_t rs = (size_t)3 * Cm;
    const float* base = qkv + (size_t)b * N * rs + (size_t)hd * 64;
    const float* qb = base;
    const float* kb = base + Cm;
    const float* vb = base + 2 * Cm;
    for (int e0 = tid; e0 < 16 * NFP * 16; e0 += 4 * blockDim.x) {     // 16 float4 per 64-float row
        float4 k4[4], v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * blockDim.x, row = e >> 4, c4 = (e & 15) * 4;
            k4[u] = make_float4(0.f, 0.f, 0.f, 0.f); v4[u] = k4[u];
            if (e < 16 * NFP * 16 && row < N) {
                k4[u] = *reinterpret_cast<const float4*>(kb + (size_t)row * rs + c4);
                v4[u] = *reinterpret_cast<const float4*>(vb + (size_t)row * rs + c4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * blockDim.x, row = e >> 4, c4 = (e & 15) * 4;
            if (e < 16 * NFP * 16) {
                *reinterpret_cast<f16x4*>(sK + row * AH_KLD + c4) = f16x4{(_Float16)k4[u].x, (_Float16)k4[u].y, (_Float16)k4[u].z, (_Float16)k4[u].w};
                sVt[(c4 + 0) * VLD + row] = (_Float16)v4[u].x;
                sVt[(c4 + 1) * VLD + row] = (_Float16)v4[u].y;
                sVt[(c4 + 2) * VLD + row] = (_Float16)v4[u].z;
                sVt[(c4 + 3) * VLD + row] = (_Float16)v4[u].w;
            }
        }
    }
    const int g = lane >> 4, li = lane & 15;
    const int qrow = min(wave * 16 + li, N - 1);
    f16x8 qh[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(qb + (size_t)qrow * rs + 32 * c + 8 * g);
        const float4 bq = *reinterpret_cast<const float4*>(qb + (size_t)qrow * rs + 32 * c + 8 * g + 4);
        qh[c] = f16x8{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)bq.x, (_Float16)bq.y, (_Float16)bq.z, (_Float16)bq.w};
    }
    __syncthreads();
    f32x4 s[NFP];
#pragma unroll
    for (int f = 0; f < NFP; ++f) {
        s[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f16x8 kv = *reinterpret_cast<const f16x8*>(sK + (16 * f + li) * AH_KLD + 32 * c + 8 * g);
            s[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kv, qh[c], s[f], 0, 0, 0);      // D[i = key 4g+r][j = query li]
        }
    }
    const float scale = 0.125f;  // 64^-0.5
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < NFP; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * f + 4 * g + r;
            const float v = key < N ? s[f][r] * scale : -INFINITY;
            s[f][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int f = 0; f < NFP; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = expf(s[f][r] - mx);
            s[f][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int F2 = 0; F2 < NFP / 2; ++F2) {
        const f16x8 ph = {(_Float16)(s[2 * F2][0] * inv), (_Float16)(s[2 * F2][1] * inv), (_Float16)(s[2 * F2][2] * inv), (_Float16)(s[2 * F2][3] * inv),
                          (_Float16)(s[2 * F2 + 1][0] * inv), (_Float16)(s[2 * F2 + 1][1] * inv), (_Float16)(s[2 * F2 + 1][2] * inv),
                          (_Float16)(s[2 * F2 + 1][3] * inv)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const _Float16* vp = sVt + (16 * e + li) * VLD + 32 * F2 + 4 * g;
            const f16x4 v0 = *reinterpret_cast<const f16x4*>(vp), v1 = *reinterpret_cast<const f16x4*>(vp + 16);
            const f16x8 vh = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[e], 0, 0, 0);          // D[i = d 16e+4g+r][j = query li]
        }
    }
    const int qo = wave * 16 + li;
    if (qo < N) {
        float* op = out + ((size_t)b * N + qo) * Cm + (size_t)hd * 64 + 4 * g;
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<float4*>(op + 16 * e) = make_float4(o[e][0], o[e][1], o[e][2], o[e][3]);
    }
}

template <int NF>
static hipError_t launch_attention_h(const float* qkv, float* out, int B, int N, int heads, hipStream_t st) {
    constexpr int NFP = (NF + 1) / 2 * 2;
    const size_t lds = ((size_t)16 * NFP * AH_KLD + (size_t)64 * (16 * NFP + 8)) * sizeof(_Float16);
    {
        hipError_t e = nd_allow_dynamic_lds((const void*)k_attention_h<NF>, lds);
        if (e != hipSuccess) return e;
    }
    int qs = 1;
    while (qs < 4 && (long)B * heads * qs < 768 && (NF + qs) / (qs + 1) >= 2) ++qs;
    if (qs > NF) qs = NF;
    const int wpq = (NF + qs - 1) / qs;
    hipLaunchKernelGGL((k_attention_h<NF>), dim3(B * heads, (NF + wpq - 1) / wpq), dim3(wpq * 64), lds, st, qkv, out, B, N, heads);
    return hipGetLastError();
}

static int attention_any(const float* qkv, float* out, int B, int N, int heads, int d, int dtype, int split_out, void* stream) {
    if (!qkv || !out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (d != 64) return nd_set_err(ND_ERR_ARG, "head dim must be 64 (got %d)", d);
    if (dtype != ND_DTYPE_F32 && dtype != ND_DTYPE_F16) return nd_set_err(ND_ERR_ARG, "unknown dtype %d", dtype);
    if (B < 1 || heads < 1 || N < 1 || N > 16 * AT_MAXF) return nd_set_err(ND_ERR_ARG, "need 1 <= N <= %d", 16 * AT_MAXF);
    hipStream_t st = (hipStream_t)stream;
    const int nf = (N + 15) / 16;
#define AT_CASE(NFV) case NFV: if (dtype == ND_DTYPE_F16) HIP_CHECK((launch_attention_h<NFV>(qkv, out, B, N, heads, st)));  \
                               else HIP_CHECK(nd_launch_attention_ring(qkv, out, B, N, heads, split_out, st)); break;
    switch (nf) {
        AT_CASE(1) AT_CASE(2) AT_CASE(3) AT_CASE(4) AT_CASE(5) AT_CASE(6) AT_CASE(7) AT_CASE(8)
        AT_CASE(9) AT_CASE(10) AT_CASE(11) AT_CASE(12) AT_CASE(13) AT_CASE(14) AT_CASE(15) AT_CASE(16)
    }
#undef AT_CASE
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

extern "C" int nd_attention(const float* qkv, float* out, int B, int N, int heads, int d, int dtype, void* stream) {
    return attention_any(qkv, out, B, N, heads, d, dtype, 0, stream);
}

// the fp32 attention with its result written as the frag32b3 image of [B*N, heads*64]: the input of the proj nd_gemm_split
extern "C" int nd_attention_split(const float* qkv, void* out_split, int B, int N, int heads, int d, void* stream) {
    return attention_any(qkv, (float*)out_split, B, N, heads, d, ND_DTYPE_F32, 1, stream);
}

// The attention core on the bf16 matrix pipe with exact fp32 products (k_attention_b9): operands = the qkv images nd_gemm_split_qkv wrote.
// out_dev: fp32 [B*N, heads*64] (out_is_split == 0) or the frag32b3 image of that matrix (the input of the proj nd_gemm_split).
extern "C" int nd_attention_images(const void* qkv_images, void* out, int out_is_split, int B, int N, int heads, void* stream) {
    if (!qkv_images || !out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (B < 1 || heads < 1 || N < 1) return nd_set_err(ND_ERR_ARG, "bad B / N / heads");
    if (!nd_qkv_images_supported(N, heads)) return nd_set_err(ND_ERR_ARG, "qkv images need N %% 4 == 0, N <= 256, heads * 64 %% 128 == 0 (N=%d, heads=%d)", N, heads);
    if (((uintptr_t)qkv_images | (uintptr_t)out) & 15) return nd_set_err(ND_ERR_ARG, "tensors must be 16-byte aligned");
    HIP_CHECK(nd_launch_attention_b9(qkv_images, (float*)out, B, N, heads, out_is_split ? 1 : 0, (hipStream_t)stream));
    return ND_OK;
}

// ---------------------------------------------------------------------------------------------
// im2col for Conv2d(k = p, stride = p): cols[(b, py, px)][c*p*p + iy*p + ix] = img[b][c][py*p+iy][px*p+ix]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_patchify(const float* __restrict__ img, float* __restrict__ cols, int B, int Cin,
                                                  int Himg, int Wimg, int p, int split_out) {
    const int gw = Wimg / p, gh = Himg / p;
    const size_t rowlen = (size_t)Cin * p * p;
    const size_t total4 = (size_t)B * gh * gw * rowlen / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const size_t tok = e / rowlen;
        const int col = (int)(e % rowlen);
        const int c = col / (p * p), iy = (col / p) % p, ix = col % p;
        const int b = (int)(tok / (gh * gw)), py = (int)(tok % (gh * gw)) / gw, px = (int)(tok % gw);
        const float4 v = *reinterpret_cast<const float4*>(img + (((size_t)b * Cin + c) * Himg + (py * p + iy)) * Wimg + px * p + ix);
        if (split_out) nd_b9_store4(reinterpret_cast<bf16x8*>(cols), (int)(rowlen >> 5), (int)tok, col, v.x, v.y, v.z, v.w);
        else *reinterpret_cast<float4*>(cols + e) = v;
    }
}

// im2col -> frag32b3, one wave per (16 tokens x 32 columns) block: a lane gathers the 8 consecutive pixels of its (token, patch row)
// -- 32 contiguous bytes of the image when p % 8 == 0 -- and the wave writes three coalesced 1 KiB planes (the element-wise form
// above scatters 8-byte pieces: 23 us per [32, 3, 224, 224] batch against 12 here).
__global__ __launch_bounds__(256) void k_patchify_split(const float* __restrict__ img, bf16x8* __restrict__ out, int B, int Cin, int Himg, int Wimg,
                                                        int p) {
    const int lane = threadIdx.x & 63;
    const long blk = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int gw = Wimg / p, gh = Himg / p, rowlen = Cin * p * p, nkb = rowlen >> 5;
    const int R = B * gh * gw;
    if (blk >= (long)((R + 15) >> 4) * nkb) return;
    const int rb = (int)(blk / nkb), kb = (int)(blk - (long)rb * nkb);
    const int tok = rb * 16 + (lane & 15), col = kb * 32 + 8 * (lane >> 4);
    float v[8];
    if (tok < R) {
        const int c = col / (p * p), iy = (col / p) % p, ix = col % p;
        const int b = tok / (gh * gw), py = (tok % (gh * gw)) / gw, px = tok % gw;
        const float* q = img + (((size_t)b * Cin + c) * Himg + (py * p + iy)) * Wimg + px * p + ix;
        const float4 u0 = *reinterpret_cast<const float4*>(q), u1 = *reinterpret_cast<const float4*>(q + 4);
        v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    bf16x8 h1, h2, h3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 a, b2, c2;
        nd_b9_split(v[e], a, b2, c2);
        h1[e] = a; h2[e] = b2; h3[e] = c2;
    }
    out[(blk * 3 + 0) * 64 + lane] = h1;
    out[(blk * 3 + 1) * 64 + lane] = h2;
    out[(blk * 3 + 2) * 64 + lane] = h3;
}

static int patchify_any(const float* img, float* cols, int B, int Cin, int Himg, int Wimg, int p, int split_out, void* stream) {
    if (!img || !cols) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (B < 1 || Cin < 1 || p < 4 || (p % 4) || Himg % p || Wimg % p || (Wimg % 4))
        return nd_set_err(ND_ERR_ARG, "patch size must be a multiple of 4 dividing the image");
    if (split_out && ((Cin * p * p) % 32)) return nd_set_err(ND_ERR_ARG, "a split (frag32b3) output needs Cin*p*p %% 32 == 0");
    const size_t total4 = (size_t)B * Cin * Himg * Wimg / 4;
    const unsigned blocks = (unsigned)((total4 + 255) / 256 > 4096 ? 4096 : (total4 + 255) / 256);
    if (split_out && (p % 8) == 0) {
        const long nb = (long)((B * (Himg / p) * (Wimg / p) + 15) / 16) * ((Cin * p * p) / 32);
        hipLaunchKernelGGL(k_patchify_split, dim3((unsigned)((nb * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, reinterpret_cast<bf16x8*>(cols), B,
                           Cin, Himg, Wimg, p);
        HIP_CHECK(hipGetLastError());
        return ND_OK;
    }
    hipLaunchKernelGGL(k_patchify, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, cols, B, Cin, Himg, Wimg, p, split_out);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

extern "C" int nd_patchify(const float* img, float* cols, int B, int Cin, int Himg, int Wimg, int p, void* stream) {
    return patchify_any(img, cols, B, Cin, Himg, Wimg, p, 0, stream);
}

// the same im2col written as the frag32b3 image of [B*(Himg/p)*(Wimg/p), Cin*p*p]: the input of the patch-embedding nd_gemm_split
extern "C" int nd_patchify_split(const float* img, void* cols_split, int B, int Cin, int Himg, int Wimg, int p, void* stream) {
    return patchify_any(img, (float*)cols_split, B, Cin, Himg, Wimg, p, 1, stream);
}
