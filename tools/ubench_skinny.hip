// ubench_skinny.hip -- design-space microbenchmark for the skinny-M weight-streaming GEMM
// (the lin2 / lin3 ConditionalLinear blocks: M = 32, K = N = 4096, G members per launch).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_skinny.hip -o tools/ubench_skinny
// Each variant computes out[g][m][n] = sum_k x[g][m][k] * w[g][n][k] (checked against variant 0),
// interleaved rounds in one process, median + min reported as GB/s of weight bytes.
#include <hip/hip_runtime.h>
#define ND_WG_TIMING
#include "../nested_diffusion_amd/csrc/nd_common.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct P {
    const float* x;    // row-major [G][M][K]
    const float* xp;   // packed     [G][M/16][K/16][64 lanes][4]
    const float* w;    // row-major [G][N][K]
    const float* wp;   // packed     [G][N/16][K/16][64][4]
    float* out;        // [G][M][N]
    int M, K, N;
};

// WPACK/XPACK: operand in MFMA fragment order (every wave load = 1 KiB contiguous)
// NF: 16-row W fragments per wave (workgroup n-tile = 16*NF); MT: 16-row x fragments; WAVES split K.
// U: chunks per pipeline stage.  NT: nontemporal W loads.  ILV: interleave accumulators in the MFMA order.
template <int MT, int NF, int WAVES, int U, bool WPACK, bool XPACK, bool NT, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64) void k_var(P p) {
    const int g = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, N = p.N, M = p.M;
    const int nch = K >> 4;
    const int ntile = blockIdx.x;  // 16*NF rows
    const float* wbase[NF];
    const float* xbase[MT];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int nf = ntile * NF + f;  // 16-row fragment index
        if (WPACK) wbase[f] = p.wp + ((size_t)g * (N / 16) + nf) * (size_t)nch * 256 + lane * 4;
        else wbase[f] = p.w + ((size_t)g * N + nf * 16 + (lane & 15)) * K + 4 * (lane >> 4);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (XPACK) xbase[mt] = p.xp + ((size_t)g * (M / 16) + mt) * (size_t)nch * 256 + lane * 4;
        else xbase[mt] = p.x + ((size_t)g * M + mt * 16 + (lane & 15)) * K + 4 * (lane >> 4);
    }
    constexpr size_t WS = WPACK ? 256 : 16;  // floats per chunk step
    constexpr size_t XS = XPACK ? 256 : 16;
    f32x4 acc[NF][MT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ngroups = nch / U;
    const int ngw = ngroups > wave ? (ngroups - wave + WAVES - 1) / WAVES : 0;
    const int glast = ngroups - 1;
    float4 wc[U][NF], xc[U][MT], wn[U][NF], xn[U][MT];
    auto LD = [&](float4 (&w)[U][NF], float4 (&x)[U][MT], int grp) {
        if (ABL == 3) grp = 0;
        const size_t c0 = (size_t)grp * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float4* a = reinterpret_cast<const float4*>(wbase[f] + (c0 + u) * WS);
                if (NT) {
                    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a));
                    w[u][f] = make_float4(v[0], v[1], v[2], v[3]);
                } else w[u][f] = *a;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (ABL == 2) x[u][mt] = make_float4(1.f, 2.f, 3.f, 4.f);
                else if (ABL == 4) x[u][mt] = *reinterpret_cast<const float4*>(xbase[mt] + (size_t)u * XS);
                else if (ABL == 5) x[u][mt] = *reinterpret_cast<const float4*>(xbase[mt] + ((size_t)(grp & 7) * U + u) * XS);
                else x[u][mt] = *reinterpret_cast<const float4*>(xbase[mt] + (c0 + u) * XS);
            }
        }
    };
    if (ngw > 0) LD(wc, xc, min(wave, glast));
    for (int i = 0; i < ngw; ++i) {
        if (ABL != 3 || i == 0) LD(wn, xn, min(wave + (i + 1) * WAVES, glast));
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const float wv = j == 0 ? wc[u][f].x : j == 1 ? wc[u][f].y : j == 2 ? wc[u][f].z : wc[u][f].w;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float xv = j == 0 ? xc[u][mt].x : j == 1 ? xc[u][mt].y : j == 2 ? xc[u][mt].z : xc[u][mt].w;
                        if (ABL == 1 || ABL == 2) { if (j == 0) acc[f][mt][0] += wv + xv; }
                        else acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[f][mt], 0, 0, 0);
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
    __shared__ float red[WAVES][NF][MT][4][64];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][f][mt][r][lane] = acc[f][mt][r];
    __syncthreads();
    for (int e = tid; e < NF * MT * 256; e += WAVES * 64) {
        const int f = e / (MT * 256), mt = (e / 256) % MT, r = (e >> 6) & 3, l = e & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w][f][mt][r][l];
        const int n = (ntile * NF + f) * 16 + 4 * (l >> 4) + r, m = mt * 16 + (l & 15);
        p.out[((size_t)g * M + m) * N + n] = s;
    }
}

// pure streaming ceiling: read all of W with 1 KiB wave loads, sum, write one float per wave
template <bool NT>
__global__ __launch_bounds__(256) void k_stream(const float* w, size_t n4, float* out) {
    const float4* p = reinterpret_cast<const float4*>(w);
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a, b, c, d;
        if (NT) {
            f32x4 t0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
            f32x4 t1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i + stride));
            f32x4 t2 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i + 2 * stride));
            f32x4 t3 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i + 3 * stride));
            s += t0[0] + t1[1] + t2[2] + t3[3];
        } else {
            a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride];
            s += a.x + b.y + c.z + d.w;
        }
    }
    for (; i < n4; i += stride) s += p[i].x;
    if (s == 123.456f) out[0] = s;
}

// Streaming-shape study: W-only reads in the loop shape of the balanced kernel.  Each wave keeps D 1-KiB loads in
// flight and issues the next D before consuming the previous D.  MAP 0: workgroup b owns a contiguous 1/gridDim of W,
// its waves interleave D-KiB pieces.  MAP 1: grid-stride (piece index = it*gridDim*WAVES + b*WAVES + wave).
// MAP 2: wave-major grid-stride (piece = (it*WAVES + wave)*gridDim + b): the chip sweeps memory front to back.
template <int WAVES, int D, int MAP, int OPT = 0>
__global__ __launch_bounds__(WAVES * 64) void k_stream2(const float* w, size_t nkib, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float big[(OPT & 1) ? 40960 : 1];
    if (OPT & 1) big[threadIdx.x] = (float)lane;
    const size_t pieces = nkib / D;                      // D-KiB pieces
    const size_t per_wg = pieces / gridDim.x;
    const size_t nit = MAP == 0 ? per_wg / WAVES : pieces / ((size_t)gridDim.x * WAVES);
    auto piece = [&](size_t it) -> size_t {
        if (MAP == 0) return (size_t)blockIdx.x * per_wg + it * WAVES + wave;
        if (MAP == 1) return (it * gridDim.x + blockIdx.x) * WAVES + wave;
        return (it * WAVES + wave) * gridDim.x + blockIdx.x;
    };
    f32x4 cur[D], nxt[D];
    float s = 0.f;
    auto LD = [&](f32x4 (&r)[D], size_t it) {
        const float* a = w + piece(it) * (size_t)D * 256 + lane * 4;
#pragma unroll
        for (int d = 0; d < D; ++d) r[d] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + d * 256));
    };
    if (nit > 0) LD(cur, 0);
    f32x4 accv[(OPT & 2) ? 10 : 1];
    if (OPT & 2) for (int q = 0; q < 10; ++q) accv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t nit_w = (OPT & 4) ? nit - (size_t)(wave & 1) * 0 + (wave == 99 ? 1 : 0) : nit;   // OPT 4: loop bound in a VGPR
    for (size_t it = 0; it < nit_w; ++it) {
        LD(nxt, it + 1 < nit ? it + 1 : it);
        if (OPT & 2) {          // ~80 dependent-free VALU ops per turn, like the ablated MFMA body
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int d = 0; d < D; ++d) { accv[2 * d][j] += cur[d][j] + 1.0f; accv[2 * d + 1][j] += cur[d][j] + 2.0f; }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) s += cur[d][d & 3];
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = nxt[d];
    }
    if (OPT & 2) for (int q = 0; q < 10; ++q) s += accv[q][0] + accv[q][1] + accv[q][2] + accv[q][3];
    if (OPT & 1) { __syncthreads(); s += big[(threadIdx.x * 7) & 1023]; }
    if (s == 123.456f) out[0] = s;
}

// Streaming + activation study: as k_stream2 MAP 0 (256 x 16 waves, D weight KiB per turn), plus X 1-KiB loads per turn
// from a small L2-resident buffer walked like the x operand (xs bytes per workgroup region).  XM: 0 = global loads issued
// after the W loads, 1 = before them, 2 = from LDS (64 KiB ring), 3 = global with two turns of W kept in flight (A/B stages).
template <int D, int X, int XM, int MF = 0, int WV = 16>
__global__ __launch_bounds__(WV * 64) void k_stream3(const float* w, size_t nkib, const float* xb, size_t xkib, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t pieces = nkib / D, per_wg = pieces / gridDim.x, nit = per_wg / WV;
    __shared__ __attribute__((aligned(16))) float xl[16384];
    if (XM == 2) {
        for (int e = threadIdx.x; e < 4096; e += WV * 64) reinterpret_cast<float4*>(xl)[e] = reinterpret_cast<const float4*>(xb)[e];
        __syncthreads();
    }
    typedef __attribute__((address_space(1))) const f32x4 gf4;
    const float* wbase = w + ((size_t)blockIdx.x * per_wg + wave) * (size_t)D * 256 + lane * 4;
    const float* xbase = xb + (size_t)(blockIdx.x % 5) * (xkib / 5) * 256 + lane * 4;     // 5 "members"
    const size_t xper = xkib / 5 / X;
    f32x4 wa[D], wb_[D], xa[X], xb_[X];
    float s = 0.f;
    auto LDW = [&](f32x4 (&r)[D], size_t it) {
        const float* a = wbase + it * WV * (size_t)D * 256;
#pragma unroll
        for (int d = 0; d < D; ++d) r[d] = __builtin_nontemporal_load((const gf4*)(a + d * 256));
    };
    auto LDX = [&](f32x4 (&r)[X], size_t it) {
        const size_t pi = (it * WV + wave) % xper;
#pragma unroll
        for (int d = 0; d < X; ++d) {
            if (XM == 2) r[d] = *reinterpret_cast<const f32x4*>(xl + (((pi * X + d) & 15) * 256) + lane * 4);
            else r[d] = *(const gf4*)(xbase + (pi * X + d) * 256);
        }
    };
    f32x4 acc[MF ? D : 1][MF ? X : 1];
    if (MF) for (int d = 0; d < D; ++d) for (int q = 0; q < X; ++q) acc[d][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto USE = [&](const f32x4 (&a)[D], const f32x4 (&x)[X]) {
        if (MF) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int d = 0; d < D; ++d)
#pragma unroll
                    for (int q = 0; q < X; ++q) acc[d][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][j], x[q][j], acc[d][q], 0, 0, 0);
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) s += a[d][0] * x[d % X][1] + a[d][2] * x[d % X][3];
        }
    };
    if (XM == 3) {
        if (nit > 0) { LDW(wa, 0); LDX(xa, 0); }
        size_t it = 0;
        for (; it + 1 < nit; it += 2) {
            LDW(wb_, it + 1); LDX(xb_, it + 1);
            __builtin_amdgcn_sched_barrier(0);
            USE(wa, xa);
            __builtin_amdgcn_sched_barrier(0);
            LDW(wa, it + 2 < nit ? it + 2 : it + 1); LDX(xa, it + 2 < nit ? it + 2 : it + 1);
            __builtin_amdgcn_sched_barrier(0);
            USE(wb_, xb_);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (it < nit) USE(wa, xa);
    } else if (XM == 4) {          // arithmetic only: operands loaded once
        LDW(wa, 0); LDX(xa, 0);
        for (size_t it = 0; it < nit; ++it) {
            USE(wa, xa);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (size_t it = 0; it < nit; ++it) {
            if (XM == 1) LDX(xa, it);
            LDW(wa, it);
            if (XM != 1) LDX(xa, it);
            __builtin_amdgcn_sched_barrier(0);
            USE(wa, xa);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (MF) for (int d = 0; d < D; ++d) for (int q = 0; q < X; ++q) s += acc[d][q][0] + acc[d][q][1] + acc[d][q][2] + acc[d][q][3];
    if (s == 123.456f) out[0] = s;
}

// k_var3's W-only loop and addressing (fragment-major, NF streams per workgroup, clamped group index, cur<-next copies),
// nothing else.  VAR 1: also carries 40 accumulator registers updated per turn like the ablated MFMA body.
template <int NF, int WAVES, int VAR>
__global__ __launch_bounds__(WAVES * 64) void k_stream5(const float* wp_, int nch, int total, float* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = blockIdx.x * NF;
    const float* wbase[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) wbase[f] = wp_ + (size_t)min(f0 + f, total - 1) * (size_t)nch * 256 + lane * 4;
    const int ngroups = nch, ngw = ngroups > wave ? (ngroups - wave + WAVES - 1) / WAVES : 0, glast = ngroups - 1;
    float4 wc[NF], wn[NF];
    f32x4 acc[NF][2];
#pragma unroll
    for (int f = 0; f < NF; ++f) { acc[f][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = acc[f][0]; }
    auto LDW = [&](float4 (&w)[NF], int grp) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wbase[f] + (size_t)grp * 256));
            w[f] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    if (ngw > 0) LDW(wc, min(wave, glast));
    for (int i = 0; i < ngw; ++i) {
        LDW(wn, min(wave + (i + 1 < ngw ? i + 1 : i) * WAVES, glast));
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            acc[f][0][0] += wc[f].x; acc[f][0][1] += wc[f].y; acc[f][0][2] += wc[f].z; acc[f][0][3] += wc[f].w;
            if (VAR == 1) { acc[f][1][0] += wc[f].x + 1.f; acc[f][1][1] += wc[f].y + 1.f; acc[f][1][2] += wc[f].z + 1.f; acc[f][1][3] += wc[f].w + 1.f; }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) wc[f] = wn[f];
    }
    float t = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) t += acc[f][0][0] + acc[f][0][1] + acc[f][0][2] + acc[f][0][3] + acc[f][1][0] + acc[f][1][3];
    if (t == 123.456f) out[0] = t;
}

// rewrites the packed x operand the way the step head does: one row m per workgroup, 16-byte pieces at a 256-byte stride
// (a 128-byte line collects pieces of 8 rows = 8 workgroups, usually on different XCDs)
__global__ __launch_bounds__(256) void k_scatter_x(const float* __restrict__ x, float* __restrict__ xp, int M, int K) {
    const int g = blockIdx.z, m = blockIdx.y, n = blockIdx.x * 1024 + threadIdx.x * 4;
    if (n >= K) return;
    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)g * M + m) * K + n);
    *reinterpret_cast<float4*>(xp + (size_t)g * M * K + nd_pk(m, n, K >> 4)) = v;
}
// the same data written as whole lines: a workgroup covers a 16-row tile x 64 columns = 4 contiguous 1-KiB blocks
__global__ __launch_bounds__(256) void k_tile_x(const float* __restrict__ x, float* __restrict__ xp, int M, int K) {
    const int g = blockIdx.z, mt = blockIdx.y, c0 = blockIdx.x * 4, tid = threadIdx.x;
    const int blk = tid >> 6, l = tid & 63, m = mt * 16 + (l & 15), n = (c0 + blk) * 16 + 4 * (l >> 4);
    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)g * M + m) * K + n);
    *reinterpret_cast<float4*>(xp + (size_t)g * M * K + ((size_t)mt * (K >> 4) + c0 + blk) * 256 + l * 4) = v;
}

__global__ void k_pack(const float* src, float* dst, int R, int K) {  // [R][K] -> [R/16][K/16][64][4]
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // float4 index in dst
    const size_t total = (size_t)R * K / 4;
    if (i >= total) return;
    const int lane = i & 63;
    const size_t blk = i >> 6;
    const int nch = K / 16;
    const size_t rt = blk / nch, c = blk % nch;
    const float4 v = *reinterpret_cast<const float4*>(src + (rt * 16 + (lane & 15)) * K + c * 16 + 4 * (lane >> 4));
    reinterpret_cast<float4*>(dst)[i] = v;
}


// Variant family 2: a workgroup owns WN*16 output columns; wave (wn, wk) streams n-fragment wn over the
// k-groups wk, wk+WK, ...  The WN waves with equal wk read the SAME x fragments (L1 hits, or LDS when XLDS).
// STAG: rotate every workgroup's k order by a blockIdx-dependent offset (spreads L2-channel hot spots).
template <int MT, int WN, int WK, int U, bool NT, bool STAG, bool XLDS>
__global__ __launch_bounds__(WN * WK * 64) void k_var2(P p) {
    const int g = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wk = wave / WN;
    const int K = p.K, N = p.N, M = p.M;
    const int nch = K >> 4;
    const int nf = blockIdx.x * WN + wn;
    const float* wbase = p.wp + ((size_t)g * (N / 16) + nf) * (size_t)nch * 256 + lane * 4;
    const float* xbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xbase[mt] = p.xp + ((size_t)g * (M / 16) + mt) * (size_t)nch * 256 + lane * 4;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ngroups = nch / U;           // assume divisible by WK
    const int ngw = ngroups / WK;
    const int rot = STAG ? (int)((blockIdx.x * 7 + blockIdx.z * 3) % ngw) : 0;
    __shared__ __attribute__((aligned(16))) float xs[XLDS ? WK : 1][2][XLDS ? MT * U * 256 : 4];
    float4 wc[U], xc[U][MT], wn_[U], xn[U][MT];
    auto grp_of = [&](int i) { int ii = i + rot; if (ii >= ngw) ii -= ngw; return wk + ii * WK; };
    auto LDW = [&](float4 (&w)[U], int grp) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* a = wbase + ((size_t)grp * U + u) * 256;
            if (NT) { f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a)); w[u] = make_float4(v[0], v[1], v[2], v[3]); }
            else w[u] = *reinterpret_cast<const float4*>(a);
        }
    };
    auto LDX = [&](float4 (&x)[U][MT], int grp) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) x[u][mt] = *reinterpret_cast<const float4*>(xbase[mt] + ((size_t)grp * U + u) * 256);
    };
    auto MM = [&](float4 (&w)[U], float4 (&x)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wv = j == 0 ? w[u].x : j == 1 ? w[u].y : j == 2 ? w[u].z : w[u].w;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float xv = j == 0 ? x[u][mt].x : j == 1 ? x[u][mt].y : j == 2 ? x[u][mt].z : x[u][mt].w;
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[mt], 0, 0, 0);
                }
            }
    };
    if (!XLDS) {
        LDW(wc, grp_of(0)); LDX(xc, grp_of(0));
        for (int i = 0; i < ngw; ++i) {
            const int gn = grp_of(i + 1 < ngw ? i + 1 : i);
            LDW(wn_, gn); LDX(xn, gn);
            MM(wc, xc);
#pragma unroll
            for (int u = 0; u < U; ++u) { wc[u] = wn_[u];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xc[u][mt] = xn[u][mt]; }
        }
    } else {
        // the WN waves of one k-split stage the x group (MT*U KiB) into LDS cooperatively: piece q of the group
        // (q = mt*U + u) is loaded by wave q % WN; double-buffered, one __syncthreads per group.
        constexpr int PIECES = MT * U;
        auto STAGE = [&](int buf, int grp) {
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                if (q % WN == wn) {
                    const int mt = q / U, u = q % U;
                    const float4 v = *reinterpret_cast<const float4*>(xbase[mt] + ((size_t)grp * U + u) * 256);
                    *reinterpret_cast<float4*>(&xs[wk][buf][q * 256 + lane * 4]) = v;
                }
            }
        };
        STAGE(0, grp_of(0));
        LDW(wc, grp_of(0));
        __syncthreads();
        for (int i = 0; i < ngw; ++i) {
            const int buf = i & 1;
            const int gn = grp_of(i + 1 < ngw ? i + 1 : i);
            LDW(wn_, gn);
            if (i + 1 < ngw) STAGE(buf ^ 1, gn);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xc[u][mt] = *reinterpret_cast<const float4*>(&xs[wk][buf][(mt * U + u) * 256 + lane * 4]);
            MM(wc, xc);
#pragma unroll
            for (int u = 0; u < U; ++u) wc[u] = wn_[u];
            __syncthreads();
        }
    }
    __shared__ float red[WK][WN][MT][4][64];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wk][wn][mt][r][lane] = acc[mt][r];
    __syncthreads();
    for (int e = tid; e < WN * MT * 256; e += WN * WK * 64) {
        const int f = e / (MT * 256), mt = (e / 256) % MT, r = (e >> 6) & 3, l = e & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WK; ++w) s += red[w][f][mt][r][l];
        const int n = (blockIdx.x * WN + f) * 16 + 4 * (l >> 4) + r, m = mt * 16 + (l & 15);
        p.out[((size_t)g * M + m) * N + n] = s;
    }
}

template <int MT, int WN, int WK, int U, bool NT, bool STAG, bool XLDS>
void launch2(P p, int G, hipStream_t st) {
    dim3 grid(p.N / (16 * WN), 1, G);
    hipLaunchKernelGGL((k_var2<MT, WN, WK, U, NT, STAG, XLDS>), grid, dim3(WN * WK * 64), 0, st, p);
}

// Variant family 3: balanced persistent partition.  The G*N/16 n-fragments are dealt contiguously, NF per
// workgroup; a wave keeps NF W fragments + MT x fragments per chunk in registers, so x is loaded once per NF
// W fragments (L1-miss traffic = W * (1 + MT/NF)).  A workgroup whose range crosses a member boundary loads
// the x of its first and of its last member ("A" and "B"; identical addresses -> L1 hit when uniform).
template <int MT, int NF, int WAVES, int U, bool NT, int ABL = 0, int ROT = 0>
__global__ __launch_bounds__(WAVES * 64) void k_var3(P p, int G) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, N = p.N, M = p.M;
    const int nch = K >> 4, nfr = N >> 4;
    const int total = G * nfr;
    const int f0 = blockIdx.x * NF;
    const float* wbase[NF];
    int gidx[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int fr = min(f0 + f, total - 1);
        gidx[f] = fr / nfr;
        wbase[f] = p.wp + (size_t)fr * (size_t)nch * 256 + lane * 4;
    }
    const int gA = gidx[0], gB = gidx[NF - 1];
    const bool uniform = gA == gB;
    const float *xA[MT], *xB[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        xA[mt] = p.xp + ((size_t)gA * (M / 16) + mt) * (size_t)nch * 256 + lane * 4;
        xB[mt] = p.xp + ((size_t)gB * (M / 16) + mt) * (size_t)nch * 256 + lane * 4;
    }
    f32x4 acc[NF][MT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ngroups = nch / U;
    const int ngw = ngroups > wave ? (ngroups - wave + WAVES - 1) / WAVES : 0;
    const int glast = ngroups - 1;
    float4 wc[U][NF], xc[U][MT], wn[U][NF], xn[U][MT];
    auto LDW = [&](float4 (&w)[U][NF], int grp) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                // ABL 4/5: chunk-major weight image (timing only: the data is not repacked) -- block (chunk c, fragment fr) at
                // (c*total + fr) KiB, so a wave's NF loads are contiguous and the chip sweeps memory front to back
                const float* a = (ABL == 4 || ABL == 5 || ABL == 7)
                    ? p.wp + ((size_t)(grp * U + u) * total + min(f0 + f, total - 1)) * 256 + lane * 4
                    : wbase[f] + ((size_t)grp * U + u) * 256;
                if (NT) { f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a)); w[u][f] = make_float4(v[0], v[1], v[2], v[3]); }
                else w[u][f] = *reinterpret_cast<const float4*>(a);
            }
    };
    // ABL == 3: x comes from a 64 KiB LDS ring filled once (timing ablation: no global x loads in the loop)
    __shared__ __attribute__((aligned(16))) float red[WAVES][NF][MT][4][64];
    float* xl = &red[0][0][0][0][0];       // the reduction buffer is free until the loop ends
    if (ABL == 3) {
        for (int e = tid; e < 16384 / 4; e += WAVES * 64) reinterpret_cast<float4*>(xl)[e] = reinterpret_cast<const float4*>(p.xp)[e];
        __syncthreads();
    }
    auto LDX = [&](float4 (&x)[U][MT], const float* const (&xb)[MT], int grp) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (ABL == 3) x[u][mt] = *reinterpret_cast<const float4*>(xl + (((grp * U + u) & 15) * MT + mt) * 256 + lane * 4);
                else x[u][mt] = *reinterpret_cast<const float4*>(xb[mt] + ((size_t)grp * U + u) * 256);
            }
    };
    if (uniform) {
        // ROT: rotate this workgroup's k order so that workgroups are at different offsets of their fragments
        const int rot = ROT == 0 ? 0 : (ROT == 1 ? (int)((blockIdx.x * 7u) % (unsigned)max(ngw, 1)) : (int)((blockIdx.x * 37u + blockIdx.x / 8u) % (unsigned)max(ngw, 1)));
        auto G_OF = [&](int i) { int ii = i + rot; if (ii >= ngw) ii -= ngw; return min(wave + ii * WAVES, glast); };
        if (ngw > 0) { LDW(wc, G_OF(0)); LDX(xc, xA, G_OF(0)); }
        for (int i = 0; i < ngw; ++i) {
            const int gn = G_OF(i + 1 < ngw ? i + 1 : i);
            LDW(wn, gn);
            if (ABL != 2 && ABL != 5 && ABL != 6 && ABL != 7) LDX(xn, xA, gn);
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
                            if (ABL == 0 || ABL == 3 || ABL == 4) acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[f][mt], 0, 0, 0);
                            else acc[f][mt][j] += wv + xv;          // every component used: the loads stay 16 bytes wide
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
        // boundary workgroup (at most G-1 of them): single-buffered, x of both members
        float4 xb[U][MT];
        for (int i = 0; i < ngw; ++i) {
            const int gn = wave + i * WAVES;
            LDW(wc, gn); LDX(xc, xA, gn); LDX(xb, xB, gn);
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
                            const float xbv = j == 0 ? xb[u][mt].x : j == 1 ? xb[u][mt].y : j == 2 ? xb[u][mt].z : xb[u][mt].w;
                            acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, useB ? xbv : xa, acc[f][mt], 0, 0, 0);
                        }
                    }
        }
    }
    if (ABL == 3) __syncthreads();
    if (ABL == 6 || ABL == 7) {
        float t = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) t += acc[f][mt][0];
        if (t == 123.456f) p.out[0] = t;
        return;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][f][mt][r][lane] = acc[f][mt][r];
    __syncthreads();
    for (int e = tid; e < NF * MT * 256; e += WAVES * 64) {
        const int f = e / (MT * 256), mt = (e / 256) % MT, r = (e >> 6) & 3, l = e & 63;
        const int fr = f0 + f;
        if (fr < total) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s += red[w][f][mt][r][l];
            const int g = fr / nfr, n = (fr % nfr) * 16 + 4 * (l >> 4) + r, m = mt * 16 + (l & 15);
            p.out[((size_t)g * M + m) * N + n] = s;
        }
    }
}

template <int MT, int NF, int WAVES, int U, bool NT, int ABL = 0, int ROT = 0>
void launch3(P p, int G, hipStream_t st) {
    const int total = G * (p.N / 16);
    dim3 grid((total + NF - 1) / NF, 1, 1);
    hipLaunchKernelGGL((k_var3<MT, NF, WAVES, U, NT, ABL, ROT>), grid, dim3(WAVES * 64), 0, st, p, G);
}

// Variant family 4: as family 3 (uniform workgroups only: benchmark G*N/16 divisible by NF), but the weight
// fragments are prefetched TWO groups ahead (3 register stages for W, 2 for x) to keep more bytes in flight.
template <int MT, int NF, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_var4(P p, int G) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, N = p.N, M = p.M;
    const int nch = K >> 4, nfr = N >> 4;
    const int total = G * nfr;
    const int f0 = blockIdx.x * NF;
    const float* wbase[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) wbase[f] = p.wp + (size_t)min(f0 + f, total - 1) * (size_t)nch * 256 + lane * 4;
    const int gA = min(f0, total - 1) / nfr;
    const float* xA[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xA[mt] = p.xp + ((size_t)gA * (M / 16) + mt) * (size_t)nch * 256 + lane * 4;
    f32x4 acc[NF][MT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ngw = nch > wave ? (nch - wave + WAVES - 1) / WAVES : 0;   // U = 1: one chunk per group
    const int glast = nch - 1;
    float4 w0[NF], w1[NF], w2[NF], x0[MT], x1[MT];
    auto LDW = [&](float4 (&w)[NF], int i) {
        const size_t c = (size_t)min(wave + i * WAVES, glast);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float* a = wbase[f] + c * 256;
            if (NT) { f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a)); w[f] = make_float4(v[0], v[1], v[2], v[3]); }
            else w[f] = *reinterpret_cast<const float4*>(a);
        }
    };
    auto LDX = [&](float4 (&x)[MT], int i) {
        const size_t c = (size_t)min(wave + i * WAVES, glast);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) x[mt] = *reinterpret_cast<const float4*>(xA[mt] + c * 256);
    };
    auto MM = [&](float4 (&w)[NF], float4 (&x)[MT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float wv = j == 0 ? w[f].x : j == 1 ? w[f].y : j == 2 ? w[f].z : w[f].w;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float xv = j == 0 ? x[mt].x : j == 1 ? x[mt].y : j == 2 ? x[mt].z : x[mt].w;
                    acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[f][mt], 0, 0, 0);
                }
            }
    };
    LDW(w0, 0); LDW(w1, 1); LDX(x0, 0);
    int i = 0;
    for (; i + 5 < ngw; i += 6) {          // 6 = lcm(3 W stages, 2 x stages)
        LDW(w2, i + 2); LDX(x1, i + 1); MM(w0, x0);
        LDW(w0, i + 3); LDX(x0, i + 2); MM(w1, x1);
        LDW(w1, i + 4); LDX(x1, i + 3); MM(w2, x0);
        LDW(w2, i + 5); LDX(x0, i + 4); MM(w0, x1);
        LDW(w0, i + 6); LDX(x1, i + 5); MM(w1, x0);
        LDW(w1, i + 7); LDX(x0, i + 6); MM(w2, x1);
    }
    for (; i < ngw; ++i) {                 // tail (ngw % 6 groups): simple reload
        LDW(w0, i); LDX(x0, i); MM(w0, x0);
    }
    __shared__ float red[WAVES][NF][MT][4][64];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][f][mt][r][lane] = acc[f][mt][r];
    __syncthreads();
    for (int e = tid; e < NF * MT * 256; e += WAVES * 64) {
        const int f = e / (MT * 256), mt = (e / 256) % MT, r = (e >> 6) & 3, l = e & 63;
        const int fr = f0 + f;
        if (fr < total) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s += red[w][f][mt][r][l];
            const int g = fr / nfr, n = (fr % nfr) * 16 + 4 * (l >> 4) + r, m = mt * 16 + (l & 15);
            p.out[((size_t)g * M + m) * N + n] = s;
        }
    }
}
template <int MT, int NF, int WAVES, bool NT>
void launch4(P p, int G, hipStream_t st) {
    const int total = G * (p.N / 16);
    dim3 grid((total + NF - 1) / NF, 1, 1);
    hipLaunchKernelGGL((k_var4<MT, NF, WAVES, NT>), grid, dim3(WAVES * 64), 0, st, p, G);
}

// ---- the library's own kernel (nd_common.hpp), driven through its launch helper ----
static SkinnyDesc* g_tab = nullptr;   // [3][G]: MODE0 packed out, MODE1 projection, MODE0 row-major out
static int g_G = 0, g_M = 0, g_K = 0, g_N = 0;
template <int MODE, int WHICH>
void launch_lib(P p, int G, hipStream_t st) {
    const SkinnyLaunch L = nd_skinny_launch<MODE>(g_K, g_N, g_M, G);
    nd_launch_skinny(L, SkinnyDesc{}, g_tab + (size_t)WHICH * g_G, G, g_M, /*t=*/3, st);
}

struct Var { const char* name; void (*launch)(P, int G, hipStream_t); };

template <int MT, int NF, int WAVES, int U, bool WP, bool XP, bool NT, int ABL = 0>
void launch(P p, int G, hipStream_t st) {
    dim3 grid(p.N / (16 * NF), 1, G);
    hipLaunchKernelGGL((k_var<MT, NF, WAVES, U, WP, XP, NT, ABL>), grid, dim3(WAVES * 64), 0, st, p);
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 5, M = 32, K = 4096, N = 4096;
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    const size_t wsz = (size_t)G * N * K, xsz = (size_t)G * M * K, osz = (size_t)G * M * N;
    float *w, *wp, *x, *xp, *out, *ref;
    CK(hipMalloc(&w, wsz * 4)); CK(hipMalloc(&wp, wsz * 4)); CK(hipMalloc(&x, xsz * 4)); CK(hipMalloc(&xp, xsz * 4));
    CK(hipMalloc(&out, osz * 4)); CK(hipMalloc(&ref, osz * 4));
    std::vector<float> hw(wsz), hx(xsz);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : hw) v = rnd() * 0.02f;
    for (auto& v : hx) v = rnd();
    CK(hipMemcpy(w, hw.data(), wsz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), xsz * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((wsz / 4 + 255) / 256)), dim3(256), 0, 0, w, wp, G * N, K);
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((xsz / 4 + 255) / 256)), dim3(256), 0, 0, x, xp, G * M, K);
    CK(hipDeviceSynchronize());
    P p{x, xp, w, wp, out, M, K, N};
    {
        g_G = G; g_M = M; g_K = K; g_N = N;
        float *scale, *shift, *pw, *part, *outp;
        const int T = 8, C = 2;
        CK(hipMalloc(&scale, (size_t)T * N * 4)); CK(hipMalloc(&shift, (size_t)T * N * 4)); CK(hipMalloc(&pw, (size_t)C * N * 4));
        CK(hipMalloc(&part, (size_t)G * M * C * (N / 16) * 4)); CK(hipMalloc(&outp, (size_t)G * M * N * 4));
        std::vector<float> ones((size_t)T * N, 1.0f);
        CK(hipMemcpy(scale, ones.data(), (size_t)T * N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(shift, 0, (size_t)T * N * 4));
        CK(hipMemcpy(pw, ones.data(), (size_t)C * N * 4, hipMemcpyHostToDevice));
        std::vector<SkinnyDesc> tab(3 * G);
        for (int g = 0; g < G; ++g) {
            const float* xg = xp + (size_t)g * M * K;
            const float* wg = wp + (size_t)g * N * K;
            tab[0 * G + g] = SkinnyDesc{xg, wg, scale, shift, outp + (size_t)g * M * N, nullptr, nullptr, K, N, C, ND_ACT_SOFTPLUS, 1};
            tab[1 * G + g] = SkinnyDesc{xg, wg, scale, shift, nullptr, pw, part + (size_t)g * M * C * (N / 16), K, N, C, ND_ACT_SOFTPLUS, 0};
            tab[2 * G + g] = SkinnyDesc{xg, wg, nullptr, nullptr, out + (size_t)g * M * N, nullptr, nullptr, K, N, C, ND_ACT_NONE, 0};
        }
        CK(hipMalloc(&g_tab, tab.size() * sizeof(SkinnyDesc)));
        CK(hipMemcpy(g_tab, tab.data(), tab.size() * sizeof(SkinnyDesc), hipMemcpyHostToDevice));
    }

    std::vector<Var> vars = {
        {"v0 rowW rowX  NF1 W4 U4      ", launch<2, 1, 4, 4, false, false, false>},
        {"v1 packW rowX NF1 W4 U4      ", launch<2, 1, 4, 4, true, false, false>},
        {"v2 packW packX NF1 W4 U4     ", launch<2, 1, 4, 4, true, true, false>},
        {"v3 packW packX NF2 W4 U2     ", launch<2, 2, 4, 2, true, true, false>},
        {"v4 packW packX NF2 W8 U2     ", launch<2, 2, 8, 2, true, true, false>},
        {"v5 packW packX NF1 W8 U4     ", launch<2, 1, 8, 4, true, true, false>},
        {"v6 packW packX NF1 W4 U4 nt  ", launch<2, 1, 4, 4, true, true, true>},
        {"v7 packW packX NF2 W4 U4     ", launch<2, 2, 4, 4, true, true, false>},
        {"v8 packW packX NF4 W4 U2     ", launch<2, 4, 4, 2, true, true, false>},
        {"v9 packW packX NF1 W4 U8     ", launch<2, 1, 4, 8, true, true, false>},
        {"v10 rowW rowX NF2 W4 U2      ", launch<2, 2, 4, 2, false, false, false>},
        {"v11 packW packX NF2 W8 U2 nt ", launch<2, 2, 8, 2, true, true, true>},
        {"v12 packW packX NF1 W16 U2   ", launch<2, 1, 16, 2, true, true, false>},
        {"a1 W8U4nt loads only (no mfma)", launch<2, 1, 8, 4, true, true, true, 1>},
        {"a2 W8U4nt W loads only       ", launch<2, 1, 8, 4, true, true, true, 2>},
        {"a3 W8U4 mfma only (no loads) ", launch<2, 1, 8, 4, true, true, true, 3>},
        {"a4 W8U4nt x from 4KiB (L1)   ", launch<2, 1, 8, 4, true, true, true, 4>},
        {"a5 W8U4nt x from 32KiB       ", launch<2, 1, 8, 4, true, true, true, 5>},
        {"v13 W8 U4 nt                 ", launch<2, 1, 8, 4, true, true, true>},
        {"v14 W8 U8 nt                 ", launch<2, 1, 8, 8, true, true, true>},
        {"v15 W16 U4 nt                ", launch<2, 1, 16, 4, true, true, true>},
        {"v16 W4 U8 nt                 ", launch<2, 1, 4, 8, true, true, true>},
        {"v17 NF2 W8 U4 nt             ", launch<2, 2, 8, 4, true, true, true>},
        {"L0 library k_skinny MODE 0   ", launch_lib<0, 0>},
        {"L1 library k_skinny MODE 1   ", launch_lib<1, 1>},
        {"r1 NF4 W16 3-stage nt (G=4)  ", launch4<2, 4, 16, true>},
        {"r2 NF4 W12 3-stage nt (G=4)  ", launch4<2, 4, 12, true>},
        {"r3 NF4 W8 3-stage nt (G=4)   ", launch4<2, 4, 8, true>},
        {"r4 NF4 W16 U1 2-stage (G=4)  ", launch3<2, 4, 16, 1, true>},
        {"t0 NF5 W16 U1 nt rot0        ", launch3<2, 5, 16, 1, true, 0, 0>},
        {"t1 NF5 W16 U1 nt rot*7       ", launch3<2, 5, 16, 1, true, 0, 1>},
        {"t2 NF5 W16 U1 nt rot*37      ", launch3<2, 5, 16, 1, true, 0, 2>},
        {"t3 NF5 W16 U1 loads-only rot ", launch3<2, 5, 16, 1, true, 1, 1>},
        {"t4 NF5 W16 U1 W-only rot     ", launch3<2, 5, 16, 1, true, 2, 1>},
        {"q1 NF5 W16 U1 nt loads only  ", launch3<2, 5, 16, 1, true, 1>},
        {"q2 NF5 W16 U1 nt W only      ", launch3<2, 5, 16, 1, true, 2>},
        {"q3 NF5 W16 U1 nt full (=p4)  ", launch3<2, 5, 16, 1, true, 0>},
        {"q4 NF5 W8 U2 nt W only       ", launch3<2, 5, 8, 2, true, 2>},
        {"q6 NF5 W16 U1 nt x from LDS  ", launch3<2, 5, 16, 1, true, 3>},
        {"q8 NF5 W16 U1 nt chunk-major ", launch3<2, 5, 16, 1, true, 4>},
        {"q12 W-only no epilogue       ", launch3<2, 5, 16, 1, true, 6>},
        {"q13 W-only chunkmaj no epilog", launch3<2, 5, 16, 1, true, 7>},
        {"q9 NF5 W16 U1 nt chunk-maj W ", launch3<2, 5, 16, 1, true, 5>},
        {"q10 NF5 W8 U2 nt chunk-major ", launch3<2, 5, 8, 2, true, 4>},
        {"q11 NF4 W16 U2 nt chunk-major", launch3<2, 4, 16, 2, true, 4>},
        {"q7 NF5 W8 U2 nt x from LDS   ", launch3<2, 5, 8, 2, true, 3>},
        {"q5 NF5 W8 U2 nt loads only   ", launch3<2, 5, 8, 2, true, 1>},
        {"p1 NF5 W8 U2 nt persistent   ", launch3<2, 5, 8, 2, true>},
        {"p2 NF5 W8 U1 nt persistent   ", launch3<2, 5, 8, 1, true>},
        {"p3 NF5 W4 U2 nt persistent   ", launch3<2, 5, 4, 2, true>},
        {"p4 NF5 W16 U1 nt persistent  ", launch3<2, 5, 16, 1, true>},
        {"p5 NF5 W8 U2 persistent      ", launch3<2, 5, 8, 2, false>},
        {"p6 NF5 W12 U1 nt persistent  ", launch3<2, 5, 12, 1, true>},
        {"p7 NF1 W8 U4 nt (=v13 form)  ", launch3<2, 1, 8, 4, true>},
        {"p8 NF5 W12 U2 nt persistent  ", launch3<2, 5, 12, 2, true>},
        {"p9 NF4 W16 U1 nt persistent  ", launch3<2, 4, 16, 1, true>},
        {"p10 NF4 W8 U2 nt persistent  ", launch3<2, 4, 8, 2, true>},
        {"p11 NF4 W12 U2 nt persistent ", launch3<2, 4, 12, 2, true>},
        {"p12 NF8 W8 U1 nt persistent  ", launch3<2, 8, 8, 1, true>},
        {"p13 NF6 W12 U1 nt persistent ", launch3<2, 6, 12, 1, true>},
        {"p14 NF3 W16 U2 nt persistent ", launch3<2, 3, 16, 2, true>},
        {"p15 NF2 W16 U2 nt persistent ", launch3<2, 2, 16, 2, true>},
        {"w1 WN1 WK8 U4 nt stag        ", launch2<2, 1, 8, 4, true, true, false>},
        {"w2 WN2 WK4 U4 nt             ", launch2<2, 2, 4, 4, true, false, false>},
        {"w3 WN4 WK2 U4 nt             ", launch2<2, 4, 2, 4, true, false, false>},
        {"w4 WN4 WK2 U4 nt stag        ", launch2<2, 4, 2, 4, true, true, false>},
        {"w5 WN4 WK4 U4 nt             ", launch2<2, 4, 4, 4, true, false, false>},
        {"w6 WN8 WK1 U4 nt             ", launch2<2, 8, 1, 4, true, false, false>},
        {"w7 WN4 WK2 U4 nt xlds        ", launch2<2, 4, 2, 4, true, false, true>},
        {"w8 WN4 WK4 U4 nt xlds        ", launch2<2, 4, 4, 4, true, false, true>},
        {"w9 WN8 WK2 U4 nt xlds        ", launch2<2, 8, 2, 4, true, false, true>},
        {"w10 WN8 WK1 U4 nt xlds       ", launch2<2, 8, 1, 4, true, false, true>},
        {"w11 WN4 WK2 U2 nt xlds       ", launch2<2, 4, 2, 2, true, false, true>},
        {"w12 WN2 WK4 U4 nt xlds       ", launch2<2, 2, 4, 4, true, false, true>},
        {"w13 WN8 WK2 U2 nt xlds stag  ", launch2<2, 8, 2, 2, true, true, true>},
    };
    if (argc > 3) {
        std::vector<Var> sel;
        sel.push_back(vars[0]);
        for (auto& v : vars) if (strstr(v.name, argv[3]) == v.name) sel.push_back(v);
        vars = sel;
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // reference = variant 0
    p.out = ref; vars[0].launch(p, G, st); CK(hipStreamSynchronize(st));
    std::vector<float> href(osz), hout(osz);
    CK(hipMemcpy(href.data(), ref, osz * 4, hipMemcpyDeviceToHost));
    p.out = out;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> times(vars.size() + 2);
    const double wbytes = (double)wsz * 4;
    for (int r = 0; r < rounds + 2; ++r) {
        for (size_t v = 0; v < vars.size(); ++v) {
            CK(hipEventRecord(e0, st));
            for (int rep = 0; rep < 4; ++rep) vars[v].launch(p, G, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) times[v].push_back(ms / 4);
            if (r == 0) {
                CK(hipMemcpy(hout.data(), out, osz * 4, hipMemcpyDeviceToHost));
                double err = 0, mx = 0;
                for (size_t i = 0; i < osz; ++i) { err = std::max(err, (double)fabsf(hout[i] - href[i])); mx = std::max(mx, (double)fabsf(href[i])); }
                printf("check %-32s max|d| = %.3e (max|ref| %.3e) %s\n", vars[v].name, err, mx, err <= 1e-4 * mx ? "ok" : ((vars[v].name[0]=='a' || vars[v].name[0]=='q' || vars[v].name[0]=='r' || vars[v].name[0]=='L' || (vars[v].name[0]=='t' && vars[v].name[1]>='3')) ? "(ablation)" : "MISMATCH"));
                CK(hipMemset(out, 0, osz * 4));
            }
        }
        for (int nt = 0; nt < 2; ++nt) {
            CK(hipEventRecord(e0, st));
            for (int rep = 0; rep < 4; ++rep) {
                if (nt) hipLaunchKernelGGL(k_stream<true>, dim3(2048), dim3(256), 0, st, w, wsz / 4, out);
                else hipLaunchKernelGGL(k_stream<false>, dim3(2048), dim3(256), 0, st, w, wsz / 4, out);
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) times[vars.size() + nt].push_back(ms / 4);
        }
    }
    if (argc > 3 && argv[3][0] == 'T') {      // per-workgroup timing of the library kernel (MODE 0)
        long long* dbg; CK(hipMalloc(&dbg, 3 * 8192 * 3 * 8)); CK(hipMemset(dbg, 0, 3 * 8192 * 3 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(nd_dbg_times), &dbg, sizeof dbg));
        for (int rep = 0; rep < 3; ++rep) { if (argv[3][1] == '1') launch_lib<1, 1>(p, G, st); else launch_lib<0, 0>(p, G, st); }
        CK(hipStreamSynchronize(st));
        std::vector<long long> h(8192 * 3);
        CK(hipMemcpy(h.data(), dbg + (size_t)(argv[3][1] == '1' ? 1 : 0) * 8192 * 3, h.size() * 8, hipMemcpyDeviceToHost));
        const SkinnyLaunch L = nd_skinny_launch<0>(K, N, M, G);
        long long tmin = 1LL << 62; for (unsigned b = 0; b < L.grid.x; ++b) tmin = std::min(tmin, h[b * 3]);
        printf("grid %u workgroups; per-WG (start, loop, total) in us relative to the first start [100 MHz clock]\n", L.grid.x);
        std::vector<std::pair<double, unsigned>> tot;
        for (unsigned b = 0; b < L.grid.x; ++b) tot.push_back({(h[b * 3 + 2] - tmin) / 100.0, b});
        std::sort(tot.begin(), tot.end());
        auto pr = [&](unsigned b) { printf("  wg %4u: start %6.2f  loop-end %6.2f  red-written %6.2f  activated %6.2f  end %6.2f\n", b, (h[b * 3] - tmin) / 100.0, (h[b * 3 + 1] - tmin) / 100.0, (h[(4096 + b) * 3] - tmin) / 100.0, (h[(4096 + b) * 3 + 1] - tmin) / 100.0, (h[b * 3 + 2] - tmin) / 100.0); };
        {   // loop-end by XCD (blockIdx % 8) and by member
            double sx[8] = {0}, sm[16] = {0}; int nx[8] = {0}, nmm[16] = {0};
            const unsigned wpm = L.grid.x / G;
            for (unsigned b = 0; b < L.grid.x; ++b) {
                const double le = (h[b * 3 + 1] - tmin) / 100.0;
                sx[b % 8] += le; nx[b % 8]++; sm[b / wpm] += le; nmm[b / wpm]++;
            }
            printf("mean loop-end by XCD:"); for (int i = 0; i < 8; ++i) printf(" %.1f", sx[i] / nx[i]); printf("\n");
            printf("mean loop-end by member:"); for (int i = 0; i < G; ++i) printf(" %.1f", sm[i] / nmm[i]); printf("\n");
        }
        printf("earliest finishers:\n"); for (int i = 0; i < 4; ++i) pr(tot[i].second);
        printf("median:\n"); pr(tot[tot.size() / 2].second);
        printf("latest finishers:\n"); for (size_t i = tot.size() - 8; i < tot.size(); ++i) pr(tot[i].second);
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'P') {      // does the predecessor kernel change the duration of the MODE 0 launch?
        auto tiny = [&]() { hipLaunchKernelGGL(k_pack, dim3(640), dim3(256), 0, st, x, xp, G * M, K); };   // ~ the step head: small, all CUs
        hipEvent_t ea[8], eb[8];
        for (int i = 0; i < 8; ++i) { CK(hipEventCreate(&ea[i])); CK(hipEventCreate(&eb[i])); }
        auto scatter = [&]() { hipLaunchKernelGGL(k_scatter_x, dim3(K / 1024, M, G), dim3(256), 0, st, x, xp, M, K); };
        auto tiled = [&]() { hipLaunchKernelGGL(k_tile_x, dim3(K / 64, M / 16, G), dim3(256), 0, st, x, xp, M, K); };
        for (int var = 0; var < 5; ++var) {
            double acc = 0; int n = 0;
            for (int r = 0; r < rounds + 2; ++r) {
                for (int i = 0; i < 8; ++i) {
                    if (var == 0) launch_lib<1, 1>(p, G, st); else if (var == 1) tiny(); else if (var == 2) { launch_lib<1, 1>(p, G, st); tiny(); }
                    else if (var == 3) { launch_lib<1, 1>(p, G, st); scatter(); } else { launch_lib<1, 1>(p, G, st); tiled(); }
                    CK(hipEventRecord(ea[i], st));
                    launch_lib<0, 0>(p, G, st);
                    CK(hipEventRecord(eb[i], st));
                }
                CK(hipStreamSynchronize(st));
                if (r >= 2) for (int i = 0; i < 8; ++i) { float ms; CK(hipEventElapsedTime(&ms, ea[i], eb[i])); acc += ms; ++n; }
            }
            printf("MODE 0 launch after %-34s: %.1f us\n", var == 0 ? "a MODE 1 launch" : var == 1 ? "a small kernel" : var == 2 ? "MODE 1 + small kernel" : var == 3 ? "MODE 1 + x rewritten in 16-B pieces" : "MODE 1 + x rewritten in whole lines", acc / n * 1e3);
        }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 's') {
        struct SV { const char* name; void (*fn)(const float*, size_t, float*, hipStream_t); };
        const size_t nkib = wsz / 256;
#define SVL(W_, D_, M_, G_) [](const float* ww, size_t nk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream2<W_, D_, M_>), dim3(G_), dim3(W_ * 64), 0, s_, ww, nk, o); }
        SV svs[] = {
            {"s 256wg x16w D5 contiguous ", SVL(16, 5, 0, 256)}, {"s 256wg x16w D5 gridstride ", SVL(16, 5, 1, 256)},
            {"s 256wg x16w D5 wave-major ", SVL(16, 5, 2, 256)}, {"s 512wg x8w  D5 contiguous ", SVL(8, 5, 0, 512)},
            {"s 1024wg x4w D5 contiguous ", SVL(4, 5, 0, 1024)}, {"s 1024wg x4w D5 gridstride ", SVL(4, 5, 1, 1024)},
            {"s 256wg x16w D10 contiguous", SVL(16, 10, 0, 256)}, {"s 256wg x16w D2 contiguous ", SVL(16, 2, 0, 256)},
            {"s 256wg x16w D2 gridstride ", SVL(16, 2, 1, 256)}, {"s 512wg x16w D5 contiguous ", SVL(16, 5, 0, 512)},
            {"s 2048wg x4w D4 gridstride ", SVL(4, 4, 1, 2048)}, {"s 256wg x8w D8 contiguous  ", SVL(8, 8, 0, 256)},
            {"s 1280wg x4w D4 contiguous ", SVL(4, 4, 0, 1280)}, {"s 256wg x16w D1 gridstride ", SVL(16, 1, 1, 256)},
            {"s5 frag-major NF5 W16        ", [](const float* ww, size_t nk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream5<5, 16, 0>), dim3(256), dim3(1024), 0, s_, ww, 256, 1280, o); }},
            {"s5 frag-major NF5 W16 +acc   ", [](const float* ww, size_t nk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream5<5, 16, 1>), dim3(256), dim3(1024), 0, s_, ww, 256, 1280, o); }},
            {"s5 frag-major NF1 W16 1280wg ", [](const float* ww, size_t nk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream5<1, 16, 0>), dim3(1280), dim3(1024), 0, s_, ww, 256, 1280, o); }},
            {"s 256wg x16w D5 contig +VALU   ", [](const float* ww, size_t nk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream2<16, 5, 0, 2>), dim3(256), dim3(1024), 0, s_, ww, nk, o); }},
            {"s 256wg x16w D5 contig +divloop", [](const float* ww, size_t nk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream2<16, 5, 0, 4>), dim3(256), dim3(1024), 0, s_, ww, nk, o); }},
            {"s 256wg x16w D5 contig +both   ", [](const float* ww, size_t nk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream2<16, 5, 0, 6>), dim3(256), dim3(1024), 0, s_, ww, nk, o); }},
            {"s 256wg x16w D5 contig +LDS160K", [](const float* ww, size_t nk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream2<16, 5, 0, 1>), dim3(256), dim3(1024), 0, s_, ww, nk, o); }},
        };
        if (argv[3][1] == 'x') {
            struct SX { const char* name; void (*fn)(const float*, size_t, const float*, size_t, float*, hipStream_t); };
            const size_t xkib = (size_t)G * M * K / 256;      // the real x operand: G members x [M][K] fp32
#define SXL(D_, X_, XM_) [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<D_, X_, XM_>), dim3(256), dim3(1024), 0, s_, ww, nk, xx, xk, o); }
            SX sxs[] = {
                {"sx D5 X2 global after W    ", SXL(5, 2, 0)}, {"sx D5 X2 global before W   ", SXL(5, 2, 1)},
                {"sx D5 X2 from LDS          ", SXL(5, 2, 2)}, {"sx D5 X2 global A/B stages ", SXL(5, 2, 3)},
                {"sx D5 X1 global after W    ", SXL(5, 1, 0)}, {"sx D10 X2 global after W   ", SXL(10, 2, 0)},
                {"sx D10 X4 global after W   ", SXL(10, 4, 0)}, {"sx D5 X4 global after W    ", SXL(5, 4, 0)},
                {"sx D5 X5 from LDS          ", SXL(5, 5, 2)},
                {"sx D5 X2 A/B + MFMA 16w    ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 3, 1, 16>), dim3(256), dim3(1024), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D5 X2 1stage + MFMA 16w ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 0, 1, 16>), dim3(256), dim3(1024), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D5 X2 A/B + MFMA 8w     ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 3, 1, 8>), dim3(256), dim3(512), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D5 X2 1stage + MFMA 8w  ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 0, 1, 8>), dim3(256), dim3(512), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D5 X2 MFMA only 16w     ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 4, 1, 16>), dim3(256), dim3(1024), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D5 X2 MFMA only 8w      ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 4, 1, 8>), dim3(256), dim3(512), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D5 X2 MFMA only 4w      ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 4, 1, 4>), dim3(256), dim3(256), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D5 X2 A/B + MFMA 4w     ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 3, 1, 4>), dim3(256), dim3(256), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D10 X2 A/B + MFMA 4w    ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<10, 2, 3, 1, 4>), dim3(256), dim3(256), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D10 X2 A/B + MFMA 8w    ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<10, 2, 3, 1, 8>), dim3(256), dim3(512), 0, s_, ww, nk, xx, xk, o); }},
                {"sx D5 X2 A/B no MFMA 8w    ", [](const float* ww, size_t nk, const float* xx, size_t xk, float* o, hipStream_t s_) { hipLaunchKernelGGL((k_stream3<5, 2, 3, 0, 8>), dim3(256), dim3(512), 0, s_, ww, nk, xx, xk, o); }},
            };
            for (auto& sv : sxs) {
                std::vector<float> tt;
                for (int r = 0; r < rounds + 2; ++r) {
                    CK(hipEventRecord(e0, st));
                    for (int rep = 0; rep < 4; ++rep) sv.fn(w, nkib, p.xp, xkib, out, st);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) tt.push_back(ms / 4);
                }
                std::sort(tt.begin(), tt.end());
                printf("%-32s median %8.1f us  min %8.1f us   %7.0f GB/s (median)\n", sv.name, tt[tt.size() / 2] * 1e3, tt[0] * 1e3, wbytes / (tt[tt.size() / 2] * 1e-3) / 1e9);
            }
        } else
        for (auto& sv : svs) {
            std::vector<float> tt;
            for (int r = 0; r < rounds + 2; ++r) {
                CK(hipEventRecord(e0, st));
                for (int rep = 0; rep < 4; ++rep) sv.fn(argv[3][1] == 'p' ? wp : w, nkib, out, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) tt.push_back(ms / 4);
            }
            std::sort(tt.begin(), tt.end());
            printf("%-32s median %8.1f us  min %8.1f us   %7.0f GB/s (median)\n", sv.name, tt[tt.size() / 2] * 1e3, tt[0] * 1e3, wbytes / (tt[tt.size() / 2] * 1e-3) / 1e9);
        }
    }
    printf("G=%d members, M=%d, K=N=%d, weight bytes per launch = %.1f MB\n", G, M, K, wbytes / 1e6);
    for (size_t v = 0; v < times.size(); ++v) {
        auto t = times[v];
        std::sort(t.begin(), t.end());
        const char* nm = v < vars.size() ? vars[v].name : (v == vars.size() ? "stream read (plain)           " : "stream read (nt)              ");
        printf("%-32s median %8.1f us  min %8.1f us   %7.0f GB/s (median)\n", nm, t[t.size() / 2] * 1e3, t[0] * 1e3, wbytes / (t[t.size() / 2] * 1e-3) / 1e9);
    }
    return 0;
}
