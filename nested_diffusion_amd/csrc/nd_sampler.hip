// nd_sampler.hip -- ensemble handle, BN/gain folding, the per-step kernels of the DDPM reverse loop
// and the hipGraph that replays a whole p_sample_loop.  gfx950 only.
//
// Reference path (file:line relative to the reference checkout):
//   diffusion/diffusion_utils.py:133-163   p_sample_loop
//   diffusion/diffusion_utils.py:54-111    p_sample / p_sample_t_1to0
//   diffusion/latent_model.py:93-105,169-184  ConditionalLinear / ConditionalModel.forward
#include "nd_common.hpp"
#include "nd_step.hpp"
#include "nd_cond_gemm.hpp"
#include "nd_b9.hpp"
#include "nd_persist.hpp"
#include "nd_rng.hpp"
#include "../../include/nested_diffusion.h"

#include <map>
#include <string>
#include <tuple>
#include <vector>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
extern "C" const char* nd_last_error(void) { return g_err.c_str(); }
#ifdef ND_WG_TIMING
// debug builds only (tools/wg_times.py): where k_skinny drops its per-workgroup clocks, 3 x 8192 x 3 int64
int nd_debug_set_wg_times_m0(void*); int nd_debug_set_wg_times_m1(void*); int nd_debug_set_wg_times_m2(void*);
extern "C" int nd_debug_set_wg_times(void* dev_ptr) {     // the k_skinny instantiations live in csrc/nd_skinny_m{0,1,2}.hip
    return (nd_debug_set_wg_times_m0(dev_ptr) | nd_debug_set_wg_times_m1(dev_ptr) | nd_debug_set_wg_times_m2(dev_ptr)) ? -1 : 0;
}
#endif
extern "C" const char* nd_version(void) { return "libnd_hip gfx950 f32 (f32-input MFMA streams + bf16x9 exact-product GEMMs and attention) r6"; }
int nd_set_err(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return nd_set_err(ND_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// eps[m, c] = sum over the n-tiles of lin3's projected partials (lin4.bias added by the caller); fixed
// reduction tree (thread-strided, wave shuffle, then waves in order) => reproducible.  All C classes
// are reduced in one pass (C is a template parameter: everything stays in registers).
template <int NT_THREADS, int C>
__device__ __forceinline__ void nd_reduce_eps(const float* __restrict__ epart, int NT, int m, float* red /* [NT_THREADS/64][C] */,
                                              float (&out)[C]) {
    const int tid = threadIdx.x;
    float s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        s[c] = 0.f;
        nd_gcf row = ND_GC(epart + ((size_t)m * C + c) * NT);
        for (int tl = tid; tl < NT; tl += NT_THREADS) s[c] += row[tl];
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[c] += __shfl_down(s[c], off, 64);
    if ((tid & 63) == 0)
#pragma unroll
        for (int c = 0; c < C; ++c) red[(tid >> 6) * C + c] = s[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NT_THREADS / 64; ++w) tot += red[w * C + c];
        out[c] = tot;
    }
}

#define ND_HEAD_INIT 0    // y = noise[0] + y_T_mean                       (diffusion_utils.py:139-140)
#define ND_HEAD_UPDATE 1  // y = posterior(y, eps(t_prev), noise[i])       (diffusion_utils.py:66-92)
#define ND_HEAD_GIVEN 2   // y = y_in (single eps_theta evaluation)

// Step head: finish the previous step (reduce eps, posterior update -> y_t), then the first
// ConditionalLinear block of this step:  h1 = softplus(A1[t] * (lin1.W [y_t, yhat]) + C1[t]) * xe
// (latent_model.py:173-177).  Grid (ceil(F/1024), M, members), 256 threads, 4 columns per thread.
// Every thread computes y_t redundantly (C values), so nothing crosses threads after the reduce; the
// table / xe / lin1 loads are issued before the reduction so their latency overlaps it.
template <int C>
__global__ __launch_bounds__(256) void k_step_head(MemberInline mi, const MemberDev* __restrict__ members, StepIO io, int mode, int i_step,
                                                   int t_prev, int t, int B, int M, int maxM, int F, int NT, int T) {
    typedef const __attribute__((address_space(4))) char* nd_cbytes;           // scalar loads from either source (see k_skinny)
    const MemberDev mb = nd_ldc<MemberDev>((members ? (nd_cbytes)(uintptr_t)members : (nd_cbytes)__builtin_amdgcn_kernarg_segment_ptr()) +
                                           (size_t)blockIdx.z * sizeof(MemberDev));
    const int z = blockIdx.z, m = blockIdx.y, b = m % B, tid = threadIdx.x;
    __shared__ float red[4 * C];
    const int n = blockIdx.x * 1024 + tid * 4;
    const bool live = n < F;
    const int nchF = F >> 4;
    constexpr int C2 = 2 * C;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), cc = a, xe = a;
    float w1[4][C2];
    if (live) {
        a = nd_ld16<false>(mb.A1 + (size_t)t * F + n);
        cc = nd_ld16<false>(mb.C1 + (size_t)t * F + n);
        xe = nd_ld16<false>(mb.xe + nd_pk(b, n, nchF));
        nd_gcf wrow = ND_GC(mb.lin1_w + (size_t)n * C2);     // 4 consecutive rows = 4*C2 contiguous floats
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < C2; ++q) w1[j][q] = wrow[j * C2 + q];
    }
    const int par_new = i_step & 1;                       // ybuf parity written by this step
    nd_gf ynew = ND_GW(mb.ybuf + ((size_t)par_new * maxM + m) * C);
    nd_gcf yold = ND_GC(mb.ybuf + ((size_t)(par_new ^ 1) * maxM + m) * C);
    float yv[C], yh[C], ym[C], zz[C], yo[C], eps[C];
    float al = 0.f, s_t = 0.f, s_tm1 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        yh[c] = ND_GC(io.yhat)[z * io.yhat_ms + (size_t)b * C + c];
        eps[c] = 0.f; ym[c] = 0.f; zz[c] = 0.f; yo[c] = 0.f;
    }
    if (mode != ND_HEAD_GIVEN) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ym[c] = ND_GC(io.ymean)[z * io.ymean_ms + (size_t)b * C + c];
            zz[c] = ND_GC(io.noise)[z * io.noise_ms + ((size_t)i_step * M + m) * C + c];
        }
    }
    if (mode == ND_HEAD_UPDATE) {
        al = ND_GC(io.alphas)[t_prev]; s_t = ND_GC(io.omabs)[t_prev]; s_tm1 = ND_GC(io.omabs)[t_prev - 1];
#pragma unroll
        for (int c = 0; c < C; ++c) yo[c] = yold[c];
        nd_reduce_eps<256, C>(mb.epart, NT, m, red, eps);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (mode == ND_HEAD_INIT) yv[c] = zz[c] + ym[c];
        else if (mode == ND_HEAD_UPDATE) yv[c] = nd_posterior(yo[c], ym[c], eps[c] + ND_GC(mb.lin4_b)[c], zz[c], al, s_t, s_tm1);
        else yv[c] = ND_GC(io.y_in)[z * io.yin_ms + (size_t)m * C + c];
        if (tid == 0 && blockIdx.x == 0) {
            ynew[c] = yv[c];
            if (io.seq_out && mode != ND_HEAD_GIVEN) ND_GW(io.seq_out)[z * io.seq_ms + ((size_t)i_step * M + m) * C + c] = yv[c];
        }
    }
    if (!live) return;
    float4 h;
    h.x = nd_head_element<C>(w1[0], yv, yh, a.x, cc.x, xe.x);
    h.y = nd_head_element<C>(w1[1], yv, yh, a.y, cc.y, xe.y);
    h.z = nd_head_element<C>(w1[2], yv, yh, a.z, cc.z, xe.z);
    h.w = nd_head_element<C>(w1[3], yv, yh, a.w, cc.w, xe.w);
    if (mb.h16 == 2)
        nd_b9_store4(reinterpret_cast<bf16x8*>(mb.h1), F >> 5, m, n, h.x, h.y, h.z, h.w);
    else if (mb.h16)
        *(__attribute__((address_space(1))) f16x4*)(reinterpret_cast<_Float16*>(mb.h1) + nd_pkh(m, n, F >> 5)) =
            f16x4{(_Float16)h.x, (_Float16)h.y, (_Float16)h.z, (_Float16)h.w};
    else
        *(__attribute__((address_space(1))) f32x4*)(mb.h1 + nd_pk(m, n, nchF)) = f32x4{h.x, h.y, h.z, h.w};
}

// Step head of the LDS-tiled blocks (M > 128 rows, h1 a frag32b3 image): k_step_head's arithmetic -- per element, and in the eps
// reduction tree: NT <= 64 partials are one value per lane of a 64-lane shuffle tree there, four values per lane of a 16-lane tree
// here, paired the same way, so both kernels return the same bits -- laid out for many rows.  A workgroup takes 16 rows x
// ND_HEAD_ROWS_COLS columns: it reduces eps and updates y for its rows ONCE (k_step_head: a workgroup per row and 1024 columns), and every lane
// computes the 8 consecutive k of one row that are its 16 bytes of an MFMA operand plane, so a wave writes whole 1 KiB planes of
// the image (k_step_head's 8-byte pieces land 256 bytes apart).
// Grid (ceil(F/ND_HEAD_ROWS_COLS), ceil(M/16), members), 256 threads; F % 32 == 0, NT <= 64.
#ifndef ND_HEAD_ROWS_COLS
#define ND_HEAD_ROWS_COLS 512          // columns per workgroup (a multiple of 128: 32 per wave and pass)
#endif
template <int C>
__global__ __launch_bounds__(256) void k_step_head_rows(MemberInline mi, const MemberDev* __restrict__ members, StepIO io, int mode,
                                                        int i_step, int t_prev, int t, int B, int M, int maxM, int F, int NT, int T) {
    typedef const __attribute__((address_space(4))) char* nd_cbytes;
    const MemberDev mb = nd_ldc<MemberDev>((members ? (nd_cbytes)(uintptr_t)members : (nd_cbytes)__builtin_amdgcn_kernarg_segment_ptr()) +
                                           (size_t)blockIdx.z * sizeof(MemberDev));
    const int z = blockIdx.z, tid = threadIdx.x, m0 = blockIdx.y * 16;
    constexpr int C2 = 2 * C;
    __shared__ float ys[16][C], yhs[16][C];
    {   // thread (r = tid / 16, j = tid % 16) holds partials j, j + 16, j + 32, j + 48 of row m0 + r
        const int r = tid >> 4, j = tid & 15, m = min(m0 + r, M - 1), b = m % B;
        const int par_new = i_step & 1;
        nd_gf ynew = ND_GW(mb.ybuf + ((size_t)par_new * maxM + m) * C);
        nd_gcf yold = ND_GC(mb.ybuf + ((size_t)(par_new ^ 1) * maxM + m) * C);
        float yv[C], yh[C], ym[C], zz[C], yo[C], eps[C];
        float al = 0.f, s_t = 0.f, s_tm1 = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            yh[c] = ND_GC(io.yhat)[z * io.yhat_ms + (size_t)b * C + c];
            eps[c] = 0.f; ym[c] = 0.f; zz[c] = 0.f; yo[c] = 0.f;
        }
        if (mode != ND_HEAD_GIVEN) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                ym[c] = ND_GC(io.ymean)[z * io.ymean_ms + (size_t)b * C + c];
                zz[c] = ND_GC(io.noise)[z * io.noise_ms + ((size_t)i_step * M + m) * C + c];
            }
        }
        if (mode == ND_HEAD_UPDATE) {
            al = ND_GC(io.alphas)[t_prev]; s_t = ND_GC(io.omabs)[t_prev]; s_tm1 = ND_GC(io.omabs)[t_prev - 1];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                yo[c] = yold[c];
                nd_gcf row = ND_GC(mb.epart + ((size_t)m * C + c) * NT);
                // nd_reduce_eps<256, C> at NT <= 64: lane i of wave 0 starts from 0.f + row[i] (0.f past NT); shuffle steps 32 and 16
                // pair (i, i+32) and then (i, i+16); steps 8..1 follow below; the other three waves add 0.f each
                float q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int tl = j + 16 * k;
                    const float v = row[min(tl, NT - 1)];
                    q[k] = tl < NT ? 0.f + v : 0.f;
                }
                float s = (q[0] + q[2]) + (q[1] + q[3]);
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) s += __shfl_down(s, off, 16);
                float tot = 0.f;
                tot += s;
                tot += 0.f; tot += 0.f; tot += 0.f;
                eps[c] = __shfl(tot, 0, 16);
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (mode == ND_HEAD_INIT) yv[c] = zz[c] + ym[c];
            else if (mode == ND_HEAD_UPDATE) yv[c] = nd_posterior(yo[c], ym[c], eps[c] + ND_GC(mb.lin4_b)[c], zz[c], al, s_t, s_tm1);
            else yv[c] = ND_GC(io.y_in)[z * io.yin_ms + (size_t)m * C + c];
            if (j == 0) {
                ys[r][c] = yv[c]; yhs[r][c] = yh[c];
                if (blockIdx.x == 0 && m0 + r < M) {
                    ynew[c] = yv[c];
                    if (io.seq_out && mode != ND_HEAD_GIVEN) ND_GW(io.seq_out)[z * io.seq_ms + ((size_t)i_step * M + m) * C + c] = yv[c];
                }
            }
        }
    }
    __syncthreads();
    // lane l of a wave: row m0 + (l & 15), the 8 columns kq = l >> 4 of each 32-column block -> its 16 bytes of the three planes
    const int lane = tid & 63, wave = tid >> 6, r = lane & 15, kq = lane >> 4;
    const int m = m0 + r, b = min(m, M - 1) % B, nkb = F >> 5, nchF = F >> 4;
    float yv[C], yh[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { yv[c] = ys[r][c]; yh[c] = yhs[r][c]; }
    bf16x8* img = reinterpret_cast<bf16x8*>(mb.h1);
#pragma unroll
    for (int it = 0; it < ND_HEAD_ROWS_COLS / 128; ++it) {
        const int kb = blockIdx.x * (ND_HEAD_ROWS_COLS / 32) + wave * (ND_HEAD_ROWS_COLS / 128) + it;
        if (kb >= nkb) break;
        const int n = kb * 32 + kq * 8;
        const float4 a0 = nd_ld16<false>(mb.A1 + (size_t)t * F + n), a1 = nd_ld16<false>(mb.A1 + (size_t)t * F + n + 4);
        const float4 c0 = nd_ld16<false>(mb.C1 + (size_t)t * F + n), c1 = nd_ld16<false>(mb.C1 + (size_t)t * F + n + 4);
        const float4 x0 = nd_ld16<false>(mb.xe + nd_pk(b, n, nchF)), x1 = nd_ld16<false>(mb.xe + nd_pk(b, n + 4, nchF));
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        nd_gcf wrow = ND_GC(mb.lin1_w + (size_t)n * C2);       // 8 consecutive rows = 8*C2 contiguous floats
        float w1[8][C2];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
            for (int q = 0; q < C2; ++q) w1[jj][q] = wrow[jj * C2 + q];
        bf16x8 p1, p2, p3;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float hv = nd_head_element<C>(w1[jj], yv, yh, av[jj], cv[jj], xv[jj]);
            __bf16 e1, e2, e3;
            nd_b9_split(hv, e1, e2, e3);
            p1[jj] = e1; p2[jj] = e2; p3[jj] = e3;
        }
        if (m < M) {
            __attribute__((address_space(1))) bf16x8* q = (__attribute__((address_space(1))) bf16x8*)(img + ((size_t)(m0 >> 4) * nkb + kb) * B9_BLOCK_UNITS + lane);
            q[0] = p1; q[64] = p2; q[128] = p3;
        }
    }
}

static void* head_rows_fn(int C) {
    switch (C) {
        case 1: return (void*)k_step_head_rows<1>;
        case 2: return (void*)k_step_head_rows<2>;
        case 3: return (void*)k_step_head_rows<3>;
        case 4: return (void*)k_step_head_rows<4>;
        case 5: return (void*)k_step_head_rows<5>;
        case 6: return (void*)k_step_head_rows<6>;
        case 7: return (void*)k_step_head_rows<7>;
        default: return (void*)k_step_head_rows<8>;
    }
}

static void* head_fn(int C) {
    switch (C) {
        case 1: return (void*)k_step_head<1>;
        case 2: return (void*)k_step_head<2>;
        case 3: return (void*)k_step_head<3>;
        case 4: return (void*)k_step_head<4>;
        case 5: return (void*)k_step_head<5>;
        case 6: return (void*)k_step_head<6>;
        case 7: return (void*)k_step_head<7>;
        default: return (void*)k_step_head<8>;
    }
}

// Last step (t = 0): y_0 = y_0_reparam (diffusion_utils.py:96-111), or plain eps output for the
// eps_theta entry point.  Grid (M, 1, members), 64 threads.
// eps_only: 0 = y_0 of the loop, 1 = eps output, 2 = one p_sample step from io.y_in with the draw in
// io.noise (t = par_cur), 3 = p_sample_t_1to0 from io.y_in.
template <int C>
__global__ __launch_bounds__(64) void k_step_final(MemberInline mi, const MemberDev* __restrict__ members, StepIO io, int eps_only, int par_cur,
                                                   int B, int M, int maxM, int NT, int T, float* eps_out, size_t eps_ms) {
    typedef const __attribute__((address_space(4))) char* nd_cbytes;           // scalar loads from either source (see k_skinny)
    const MemberDev mb = nd_ldc<MemberDev>((members ? (nd_cbytes)(uintptr_t)members : (nd_cbytes)__builtin_amdgcn_kernarg_segment_ptr()) +
                                           (size_t)blockIdx.z * sizeof(MemberDev));
    const int z = blockIdx.z, m = blockIdx.x, b = m % B;
    __shared__ float red[C];
    float epsv[C];
    nd_reduce_eps<64, C>(mb.epart, NT, m, red, epsv);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float eps = epsv[c] + mb.lin4_b[c];
        if (threadIdx.x == 0) {
            if (eps_only == 1) {
                eps_out[z * eps_ms + (size_t)m * C + c] = eps;
            } else if (eps_only >= 2) {
                const float y = io.y_in[(size_t)m * C + c];
                const float ymean = io.ymean[(size_t)b * C + c];
                const int t = par_cur;
                eps_out[(size_t)m * C + c] = eps_only == 2
                    ? nd_posterior(y, ymean, eps, io.noise[(size_t)m * C + c], io.alphas[t], io.omabs[t], io.omabs[t - 1])
                    : nd_y0_reparam(y, ymean, eps, io.omabs[0]);
            } else {
                const float y = mb.ybuf[((size_t)par_cur * maxM + m) * C + c];
                const float ymean = io.ymean[z * io.ymean_ms + (size_t)b * C + c];
                const float y0 = nd_y0_reparam(y, ymean, eps, io.omabs[0]);
                io.y0_out[z * io.y0_ms + (size_t)m * C + c] = y0;
                if (io.seq_out) io.seq_out[z * io.seq_ms + ((size_t)T * M + m) * C + c] = y0;
            }
        }
    }
}

static void* final_fn(int C) {
    switch (C) {
        case 1: return (void*)k_step_final<1>;
        case 2: return (void*)k_step_final<2>;
        case 3: return (void*)k_step_final<3>;
        case 4: return (void*)k_step_final<4>;
        case 5: return (void*)k_step_final<5>;
        case 6: return (void*)k_step_final<6>;
        case 7: return (void*)k_step_final<7>;
        default: return (void*)k_step_final<8>;
    }
}

// ---- one-time folds, computed in fp64 and rounded once --------------------------
// eval BatchNorm1d: BN(u) = s*u + o, s = w / sqrt(var + 1e-5), o = beta - mean*s
// Linear bias folded: BN(W h + b) = s*(W h) + (s*b + o)
__global__ void k_fold_bn(float* scale, float* shift, const float* lin_b, const float* bn_w, const float* bn_b,
                          const float* bn_mean, const float* bn_var, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double s = (double)bn_w[n] / sqrt((double)bn_var[n] + 1e-5);
    const double o = (double)bn_b[n] - (double)bn_mean[n] * s;
    scale[n] = (float)s;
    shift[n] = (float)(s * (double)lin_b[n] + o);
}
// ConditionalLinear + BN: BN(g_t * (W h + b)) = (s g_t) * (W h) + (s g_t b + o)
__global__ void k_fold_steps(float* A, float* Cc, const float* emb, const float* lin_b, const float* bn_w, const float* bn_b,
                             const float* bn_mean, const float* bn_var, int T, int N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * N) return;
    const int n = (int)(i % N);
    const double s = (double)bn_w[n] / sqrt((double)bn_var[n] + 1e-5);
    const double o = (double)bn_b[n] - (double)bn_mean[n] * s;
    const double g = (double)emb[i];
    A[i] = (float)(s * g);
    Cc[i] = (float)(s * g * (double)lin_b[n] + o);
}

// ---------------------------------------------------------------------------------------------
// host handle
// ---------------------------------------------------------------------------------------------
#define ND_KEEP_BYTES 208.0e6     // weights kept Infinity-Cache resident across steps (see nd_load_member)
// L_LIN2S / L_LIN3S: the two step blocks with frag32b3 operands (bf16 matrix pipe, exact fp32 products: rows > 128 only)
enum { L_ENC0 = 0, L_ENC1 = 1, L_ENC2 = 2, L_LIN2 = 3, L_LIN3 = 4, L_LIN2S = 5, L_LIN3S = 6, L_COUNT = 7 };

struct MemberHost {
    bool loaded = false;
    float *sc0, *sh0, *sc1, *sh1, *sc2, *sh2;
    float *A[3], *Cc[3];
    float *w_enc0, *w_enc3, *w_enc6, *w_lin2, *w_lin3;   // frag16-packed weights
    float *w_lin1, *w_lin4, *b_lin4;                      // small row-major copies
    float *e0, *e1, *xe, *ybuf, *h1, *h2, *epart, *splitk;
    void *w_lin2s = nullptr, *w_lin3s = nullptr, *h1s = nullptr, *h2s = nullptr;   // frag32b3 images (handles with max_rows > 128, fp32)
    bool h_split = false;        // which copies of h1 / h2 the LAST launch over this member wrote: the frag32b3 images (true) or h1 / h2 (nd_member_buffer)
};

struct GraphKey {
    int m0, nm, B, mc, T;
    const void *yhat, *ymean, *noise, *y0, *seq;
    bool operator<(const GraphKey& o) const {
        return std::tie(m0, nm, B, mc, T, yhat, ymean, noise, y0, seq) <
               std::tie(o.m0, o.nm, o.B, o.mc, o.T, o.yhat, o.ymean, o.noise, o.y0, o.seq);
    }
};

int nd_guiding_prediction_first(nd_cond c, const float* images, float* logits_out, float* yhat_out, int B, int n_used, void* stream);
unsigned long long nd_cond_serial(nd_cond c);     // nd_conditioner.hip: changes with every change of the conditioner's state

struct BatchKey {
    unsigned long long cond;          // the conditioner's serial, not its address (addresses are reused after nd_cond_destroy)
    const void *images, *noise, *samples, *prob, *vote, *probs, *yhat;
    int B, mc, T;
    unsigned temp_bits;
    bool profiling;
    bool operator<(const BatchKey& o) const {
        return std::tie(cond, images, noise, samples, prob, vote, probs, yhat, B, mc, T, temp_bits, profiling) <
               std::tie(o.cond, o.images, o.noise, o.samples, o.prob, o.vote, o.probs, o.yhat, o.B, o.mc, o.T, o.temp_bits, o.profiling);
    }
};

struct nd_handle_s {
    nd_config cfg{};
    char* ws = nullptr;
    size_t ws_bytes = 0;
    std::vector<MemberHost> members;
    MemberDev* members_dev = nullptr;      // [K]
    SkinnyDesc* descs_dev = nullptr;       // [L_COUNT][K]
    std::vector<SkinnyDesc> descs_host;    // the same table on the host: the step launches pass their rows by value (SkinnyInline)
    std::vector<MemberDev> members_host;   // ... and the step head / final kernels theirs (MemberInline)
    SkinnyDesc* spk_dev = nullptr;         // [K]   encoder_x.0 as split-K partial sums (MODE 2)
    SplitKEpiDesc* spke_dev = nullptr;     // [K]
    float *alphas = nullptr, *omabs = nullptr;
    float* xpack = nullptr;                // frag16 [maxB][D] image batch shared by all members
    float* tile_ws = nullptr;              // k-slab accumulators of k_cond_gemm's split tail (large-M steps only)
    bool b9 = false;                       // the large-M step blocks run on the bf16 matrix pipe (frag32b3 copies of lin2 / lin3 exist)
    unsigned long long* rng_state = nullptr;   // {seed, batch counter | first image << 32} of the in-library noise (nd_seed)
    unsigned* input_seq = nullptr;             // calls of nd_predict_batch whose inputs have been consumed (device counter behind input_flag)
    unsigned* input_flag = nullptr;            // caller's host-visible word that receives that count (nd_set_input_flag), or null
    unsigned* persist_bar = nullptr;           // barrier block of the one-launch loop (nd_persist.hpp): per-member arrival counters + sticky error word
    int persist_mode = 0;                      // ND_PERSIST: 0 = per-step kernels (hipGraph form), 1 = one launch per p_sample_loop where nd_persist_plan allows
    int persist_skew_ticks = 0;                // ND_PERSIST_SKEW_US: start offset between consecutive members (100 MHz ticks)
    bool persist_last = false;                 // the most recently recorded / enqueued loop took the one-launch form
    float* noise_ws = nullptr;             // [K][T][max_rows][C] draws of the in-library noise (noise_dev == NULL)
    float* logits_ws = nullptr;            // [K][max_batch][C] guiding-prediction logits of nd_predict_batch
    int sched_T = 0;
    int NT = 0, S0 = 0;
    bool enc_splitk = false;
    int half = 0;                          // cfg.operand_dtype == ND_DTYPE_F16
    std::map<GraphKey, hipGraphExec_t> graphs;
    std::map<BatchKey, hipGraphExec_t> batch_graphs;   // nd_predict_batch: the whole hot path of a batch
    hipStream_t capture_stream = nullptr;              // only ever used to RECORD batch graphs, never to run anything
    int encoded_B = -1;
    bool profiling = false;
    std::vector<hipEvent_t> probe_events;   // 4 per probed step i: e0 | head(i) | e1 | lin2, lin3 (i) | e2 | head, lin2, lin3 (i+1) | e3
    int probe_steps = 0;
    int probe_nodes = 0;             // event-record nodes in the most recently built graph (nd_profile_probe_nodes)
};

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
    char* base; size_t off = 0;
    template <typename T> T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off = al256(off + count * sizeof(T));
        return p;
    }
};

static void carve(nd_handle_s* h, char* base, size_t* total) {
    const nd_config& c = h->cfg;
    const size_t K = c.n_members, H = c.hidden_dim, F = c.feature_dim, T = c.n_steps, C = c.y_dim, D = c.data_dim;
    const size_t mB = c.max_batch, mM = c.max_rows;
    const size_t pB = ((mB + 15) / 16) * 16, pM = ((mM + 15) / 16) * 16;   // rows padded to whole 16-row tiles
    h->NT = (int)((F + 15) / 16);
    h->half = c.operand_dtype == ND_DTYPE_F16;
    const size_t wdiv = h->half ? 2 : 1;                                     // packed weights: floats -> halfs
    h->enc_splitk = nd_use_splitk(c.data_dim);
    h->S0 = h->enc_splitk ? nd_skinny_launch<2>(c.data_dim, c.hidden_dim, c.max_batch, 1, h->half).S : 0;
    Carver cv{base};
    h->members_dev = cv.take<MemberDev>(K);
    h->descs_dev = cv.take<SkinnyDesc>(L_COUNT * K);
    h->spk_dev = cv.take<SkinnyDesc>(K);
    h->spke_dev = cv.take<SplitKEpiDesc>(K);
    h->alphas = cv.take<float>(T);
    h->omabs = cv.take<float>(T);
    h->rng_state = cv.take<unsigned long long>(2);
    h->persist_bar = cv.take<unsigned>(ND_PERSIST_BAR_WORDS);
    h->input_seq = cv.take<unsigned>(1);
    h->noise_ws = cv.take<float>(K * T * mM * C);
    h->logits_ws = cv.take<float>(K * mB * C);
    h->xpack = cv.take<float>(pB * D);
    h->tile_ws = cv.take<float>(nd_cond_gemm_wanted((int)mM, h->half) ? (size_t)CG_MAX_SLABS * CG_T * CG_T : 1);
    // more than 128 rows per member and fp32: the step blocks are MFMA-bound GEMMs; they run on the bf16 matrix pipe with exact
    // fp32 products when the depth allows (F % 32 == 0), which takes a frag32b3 copy of lin2 / lin3 (1.5 x their fp32 size)
    h->b9 = nd_cond_gemm_wanted((int)mM, h->half) && (F % 32) == 0 && !getenv("ND_STEP_F32_MFMA");
    for (size_t k = 0; k < K; ++k) {
        MemberHost& m = h->members[k];
        m.sc0 = cv.take<float>(H); m.sh0 = cv.take<float>(H);
        m.sc1 = cv.take<float>(H); m.sh1 = cv.take<float>(H);
        m.sc2 = cv.take<float>(F); m.sh2 = cv.take<float>(F);
        for (int l = 0; l < 3; ++l) { m.A[l] = cv.take<float>(T * F); m.Cc[l] = cv.take<float>(T * F); }
        m.w_enc0 = cv.take<float>(H * D / wdiv); m.w_enc3 = cv.take<float>(H * H / wdiv); m.w_enc6 = cv.take<float>(F * H / wdiv);
        m.w_lin2 = cv.take<float>(F * F / wdiv); m.w_lin3 = cv.take<float>(F * F / wdiv);
        m.w_lin1 = cv.take<float>(F * 2 * C); m.w_lin4 = cv.take<float>(C * F); m.b_lin4 = cv.take<float>(C);
        m.e0 = cv.take<float>(pB * H); m.e1 = cv.take<float>(pB * H); m.xe = cv.take<float>(pB * F);
        m.ybuf = cv.take<float>(2 * mM * C);
        m.h1 = cv.take<float>(pM * F); m.h2 = cv.take<float>(pM * F);
        if (h->b9) {
            m.w_lin2s = cv.take<char>(nd_b9_bytes((int)F, (int)F)); m.w_lin3s = cv.take<char>(nd_b9_bytes((int)F, (int)F));
            m.h1s = cv.take<char>(nd_b9_bytes((int)pM, (int)F)); m.h2s = cv.take<char>(nd_b9_bytes((int)pM, (int)F));
        }
        // eps partials per (row, class): F/16 from k_skinny, 2 per 128-column tile from k_cond_gemm (more than F/16 when F = 16)
        const size_t ntl_max = nd_cond_gemm_wanted((int)mM, h->half) ? (size_t)nd_cond_gemm_plan((int)F, (int)F, (int)mM, 1, h->half).ntl : 0;
        m.epart = cv.take<float>(((size_t)h->NT > ntl_max ? (size_t)h->NT : ntl_max) * mM * C);
        // split-K slabs: sized for the deepest split any (B <= max_batch, member count) launch can pick
        m.splitk = cv.take<float>(h->enc_splitk ? (size_t)(D / 16 / 64 + 1) * pB * H : 1);
    }
    *total = cv.off;
}

static int check_cfg(const nd_config* c) {
    if (!c) return nd_set_err(ND_ERR_ARG, "cfg is NULL");
    if (c->y_dim < 1 || c->y_dim > ND_MAX_C) return nd_set_err(ND_ERR_ARG, "y_dim must be in [1,%d]", ND_MAX_C);
    if (c->data_dim < 16 || c->data_dim % 16) return nd_set_err(ND_ERR_ARG, "data_dim must be a positive multiple of 16");
    if (c->hidden_dim < 16 || c->hidden_dim % 16) return nd_set_err(ND_ERR_ARG, "hidden_dim must be a positive multiple of 16");
    if (c->feature_dim < 16 || c->feature_dim % 16) return nd_set_err(ND_ERR_ARG, "feature_dim must be a positive multiple of 16");
    if (c->n_steps < 1) return nd_set_err(ND_ERR_ARG, "n_steps must be >= 1");
    if (c->operand_dtype != ND_DTYPE_F32 && c->operand_dtype != ND_DTYPE_F16)
        return nd_set_err(ND_ERR_ARG, "operand_dtype must be ND_DTYPE_F32 or ND_DTYPE_F16");
    if (c->operand_dtype == ND_DTYPE_F16 && ((c->data_dim | c->hidden_dim | c->feature_dim) % 32))
        return nd_set_err(ND_ERR_ARG, "fp16 operands need data_dim, hidden_dim and feature_dim to be multiples of 32");
    if (c->n_members < 1 || c->n_members > 255 || c->max_batch < 1 || c->max_rows < c->max_batch)
        return nd_set_err(ND_ERR_ARG, "n_members (1..255) / max_batch / max_rows invalid");
    return ND_OK;
}

extern "C" size_t nd_workspace_bytes(const nd_config* cfg) {
    if (check_cfg(cfg) != ND_OK) return 0;
    nd_handle_s tmp;
    tmp.cfg = *cfg;
    tmp.members.resize(cfg->n_members);
    size_t total = 0;
    carve(&tmp, nullptr, &total);
    return total;
}

extern "C" int nd_create(const nd_config* cfg, nd_handle* out) {
    if (!out) return nd_set_err(ND_ERR_ARG, "out is NULL");
    int rc = check_cfg(cfg);
    if (rc != ND_OK) return rc;
    nd_handle_s* h = new nd_handle_s();
    h->cfg = *cfg;
    h->members.resize(cfg->n_members);
    if (const char* e = getenv("ND_PERSIST")) h->persist_mode = atoi(e);
    if (const char* e = getenv("ND_PERSIST_SKEW_US")) h->persist_skew_ticks = (int)(atof(e) * 100.0);
    *out = h;
    return ND_OK;
}

static void drop_graphs(nd_handle_s* h) {
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
    for (auto& kv : h->batch_graphs) (void)hipGraphExecDestroy(kv.second);
    h->batch_graphs.clear();
}

extern "C" int nd_destroy(nd_handle h) {
    if (!h) return ND_OK;
    drop_graphs(h);
    for (hipEvent_t e : h->probe_events) (void)hipEventDestroy(e);
    if (h->capture_stream) (void)hipStreamDestroy(h->capture_stream);
    delete h;
    return ND_OK;
}

extern "C" int nd_bind_workspace(nd_handle h, void* ws, size_t bytes) {
    if (!h || !ws) return nd_set_err(ND_ERR_ARG, "handle/workspace is NULL");
    if ((uintptr_t)ws & 255) return nd_set_err(ND_ERR_ARG, "workspace must be 256-byte aligned");
    size_t need = 0;
    carve(h, (char*)ws, &need);
    if (bytes < need) return nd_set_err(ND_ERR_ARG, "workspace too small: %zu < %zu", bytes, need);
    h->ws = (char*)ws;
    h->ws_bytes = bytes;
    drop_graphs(h);
    for (auto& m : h->members) m.loaded = false;
    // activations: padded tile rows must hold finite values before the first kernel reads them
    HIP_CHECK(hipMemset(h->xpack, 0, (size_t)((char*)h->members[0].sc0 - (char*)h->xpack)));
    for (auto& m : h->members) HIP_CHECK(hipMemset(m.e0, 0, (size_t)((char*)m.splitk - (char*)m.e0)));
    HIP_CHECK(hipMemset(h->rng_state, 0, 2 * sizeof(unsigned long long)));
    HIP_CHECK(hipMemset(h->persist_bar, 0, ND_PERSIST_BAR_WORDS * sizeof(unsigned)));
    HIP_CHECK(hipMemset(h->input_seq, 0, sizeof(unsigned)));
    return ND_OK;
}

extern "C" int nd_seed(nd_handle h, uint64_t seed, uint32_t first_image) {
    if (!h || !h->ws) return nd_set_err(ND_ERR_STATE, "workspace not bound");
    const unsigned long long st[2] = {(unsigned long long)seed, (unsigned long long)first_image << 32};   // batch counter 0
    HIP_CHECK(hipDeviceSynchronize());          // no sampling graph may be reading the state while it is replaced
    HIP_CHECK(hipMemcpy(h->rng_state, st, sizeof st, hipMemcpyHostToDevice));
    return ND_OK;
}

extern "C" int nd_set_schedule(nd_handle h, const float* alphas_dev, const float* omabs_dev, int T, void* stream) {
    if (!h || !h->ws) return nd_set_err(ND_ERR_STATE, "workspace not bound");
    if (!alphas_dev || !omabs_dev) return nd_set_err(ND_ERR_ARG, "NULL schedule");
    if (T < 1 || T > h->cfg.n_steps) return nd_set_err(ND_ERR_ARG, "T=%d outside [1,%d]", T, h->cfg.n_steps);
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipMemcpyAsync(h->alphas, alphas_dev, sizeof(float) * T, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpyAsync(h->omabs, omabs_dev, sizeof(float) * T, hipMemcpyDeviceToDevice, st));
    h->sched_T = T;
    return ND_OK;
}

static void launch_pack(const float* src, float* dst, int R, int K, int half, hipStream_t st) {
    const size_t n4 = (size_t)((R + 15) / 16) * 16 * K / (half ? 8 : 4);
    const size_t want = (n4 + 255) / 256;
    const dim3 grid((unsigned)(want > 8192 ? 8192 : want));
    if (half) hipLaunchKernelGGL(k_pack_rows_h, grid, dim3(256), 0, st, src, reinterpret_cast<_Float16*>(dst), R, K);
    else hipLaunchKernelGGL(k_pack_rows, grid, dim3(256), 0, st, src, dst, R, K);
}

extern "C" int nd_load_member(nd_handle h, int k, const nd_member_weights* w, void* stream) {
    if (!h || !h->ws) return nd_set_err(ND_ERR_STATE, "workspace not bound");
    if (k < 0 || k >= h->cfg.n_members) return nd_set_err(ND_ERR_ARG, "member %d out of range", k);
    if (!w) return nd_set_err(ND_ERR_ARG, "weights is NULL");
    const void* const* pp = reinterpret_cast<const void* const*>(w);
    for (size_t i = 0; i < sizeof(nd_member_weights) / sizeof(void*); ++i)
        if (!pp[i]) return nd_set_err(ND_ERR_ARG, "nd_member_weights pointer #%zu is NULL", i);
    hipStream_t st = (hipStream_t)stream;
    const nd_config& c = h->cfg;
    const int H = c.hidden_dim, F = c.feature_dim, T = c.n_steps, C = c.y_dim, D = c.data_dim;
    MemberHost& m = h->members[k];
    auto g1 = [](int n) { return dim3((n + 255) / 256); };
    hipLaunchKernelGGL(k_fold_bn, g1(H), dim3(256), 0, st, m.sc0, m.sh0, w->enc0_b, w->bn0_w, w->bn0_b, w->bn0_mean, w->bn0_var, H);
    hipLaunchKernelGGL(k_fold_bn, g1(H), dim3(256), 0, st, m.sc1, m.sh1, w->enc3_b, w->bn1_w, w->bn1_b, w->bn1_mean, w->bn1_var, H);
    hipLaunchKernelGGL(k_fold_bn, g1(F), dim3(256), 0, st, m.sc2, m.sh2, w->enc6_b, w->norm_w, w->norm_b, w->norm_mean, w->norm_var, F);
    const float* embs[3] = {w->emb1, w->emb2, w->emb3};
    const float* lb[3] = {w->lin1_b, w->lin2_b, w->lin3_b};
    const float* bw[3] = {w->un1_w, w->un2_w, w->un3_w};
    const float* bb[3] = {w->un1_b, w->un2_b, w->un3_b};
    const float* bm[3] = {w->un1_mean, w->un2_mean, w->un3_mean};
    const float* bv[3] = {w->un1_var, w->un2_var, w->un3_var};
    const size_t tot = (size_t)T * F;
    for (int l = 0; l < 3; ++l)
        hipLaunchKernelGGL(k_fold_steps, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m.A[l], m.Cc[l], embs[l], lb[l],
                           bw[l], bb[l], bm[l], bv[l], T, F);
    // weights -> frag16 in the workspace; the raw tensors are not referenced after this call returns
    launch_pack(w->enc0_w, m.w_enc0, H, D, h->half, st);
    launch_pack(w->enc3_w, m.w_enc3, H, H, h->half, st);
    launch_pack(w->enc6_w, m.w_enc6, F, H, h->half, st);
    launch_pack(w->lin2_w, m.w_lin2, F, F, h->half, st);
    launch_pack(w->lin3_w, m.w_lin3, F, F, h->half, st);
    if (h->b9) {
        int rc = nd_split_rows(w->lin2_w, m.w_lin2s, F, F, st);
        if (rc == ND_OK) rc = nd_split_rows(w->lin3_w, m.w_lin3s, F, F, st);
        if (rc != ND_OK) return rc;
        HIP_CHECK(nd_cond_gemm_b9_prepare());
    }
    HIP_CHECK(hipMemcpyAsync(m.w_lin1, w->lin1_w, sizeof(float) * F * 2 * C, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpyAsync(m.w_lin4, w->lin4_w, sizeof(float) * C * F, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpyAsync(m.b_lin4, w->lin4_b, sizeof(float) * C, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipGetLastError());

    MemberDev md{m.w_lin1, m.b_lin4, m.A[0], m.Cc[0], m.xe, m.h1, m.ybuf, m.epart, h->half};
    const int opk = h->half ? 2 : 1;     // layout of an activation that feeds the next GEMM
    SkinnyDesc ds[L_COUNT];
    ds[L_ENC0] = SkinnyDesc{h->xpack, m.w_enc0, m.sc0, m.sh0, m.e0, nullptr, nullptr, D, H, C, ND_ACT_SOFTPLUS, opk};
    ds[L_ENC1] = SkinnyDesc{m.e0, m.w_enc3, m.sc1, m.sh1, m.e1, nullptr, nullptr, H, H, C, ND_ACT_SOFTPLUS, opk};
    ds[L_ENC2] = SkinnyDesc{m.e1, m.w_enc6, m.sc2, m.sh2, m.xe, nullptr, nullptr, H, F, C, ND_ACT_NONE, 1};
    ds[L_LIN2] = SkinnyDesc{m.h1, m.w_lin2, m.A[1], m.Cc[1], m.h2, nullptr, nullptr, F, F, C, ND_ACT_SOFTPLUS, opk};
    ds[L_LIN3] = SkinnyDesc{m.h2, m.w_lin3, m.A[2], m.Cc[2], nullptr, m.w_lin4, m.epart, F, F, C, ND_ACT_SOFTPLUS, 0};
    // each block's stream ends by touching the first lines of the next launch's stream (SkinnyDesc::pf): lin2 -> lin3 of the same step,
    // lin3 -> lin2 of the next one.  ND_NO_XPREFETCH=1: off (the A/B switch).
    if (!getenv("ND_NO_XPREFETCH")) { ds[L_LIN2].pf = m.w_lin3; ds[L_LIN3].pf = m.w_lin2; }
    ds[L_LIN2S] = ds[L_LIN2]; ds[L_LIN3S] = ds[L_LIN3];
    if (h->b9) {   // the same two blocks on frag32b3 operands: h1s -> lin2 -> h2s (out_packed 3) -> lin3 + lin4 -> eps partials
        ds[L_LIN2S].x = (const float*)m.h1s; ds[L_LIN2S].w = (const float*)m.w_lin2s; ds[L_LIN2S].out = (float*)m.h2s; ds[L_LIN2S].out_packed = 3;
        ds[L_LIN3S].x = (const float*)m.h2s; ds[L_LIN3S].w = (const float*)m.w_lin3s;
    }
    {
        // INFINITY-CACHE RESIDENCY.  A step streams 2 K F^2 weights (671 MB at K = 5, fp32), far more than the 256 MiB Infinity
        // Cache, so the step kernels read W with nontemporal loads -- nothing survives to the next step.  Reading the first
        // ND_KEEP_BYTES of them (lin2 of member 0, 1, .. then lin3) with default-policy loads instead keeps exactly those resident
        // from step to step while the nontemporal rest streams past them.  Measured (K = 5, T = 100, B = 32; sampler ms):
        //   fp32: none 13.43 | 1+1 matrices 12.57 | 3+0 12.36 | 4+0 12.58 | 3+1 12.63 | all 14.44      (a matrix = 67 MB)
        //   fp16: none  7.67 | 5+0 7.11 | 5+1 6.88 | 5+2 6.96 | all 7.91                                 (a matrix = 34 MB)
        // i.e. ~200 MB is what stays (the rest of the cache turns over with activations, tables and the streamed lines).
        const double mat = (double)F * F * (h->half ? 2.0 : 4.0);
        const char* keep_mb = getenv("ND_KEEP_MB");                       // experiments: another residency budget (MB)
        const int n_keep = (int)((keep_mb ? atof(keep_mb) * 1e6 : ND_KEEP_BYTES) / mat);
        ds[L_LIN2].keep = k < n_keep;
        ds[L_LIN3].keep = c.n_members + k < n_keep;
        if (const char* plan = getenv("ND_KEEP_PLAN")) {                  // experiments: "a,b" = lin2 of members < a, lin3 of members < b
            int a = 0, b = 0;
            if (sscanf(plan, "%d,%d", &a, &b) == 2) { ds[L_LIN2].keep = k < a; ds[L_LIN3].keep = k < b; }
        }
    }
    // small synchronous H2D copies: load time only, never on the sampling path.  The sync also means the
    // caller may release its raw weight tensors as soon as this function returns.
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipMemcpy(h->members_dev + k, &md, sizeof md, hipMemcpyHostToDevice));
    h->members_host.resize(c.n_members);
    h->descs_host.resize((size_t)L_COUNT * c.n_members);
    h->members_host[k] = md;
    for (int l = 0; l < L_COUNT; ++l) {
        h->descs_host[(size_t)l * c.n_members + k] = ds[l];
        HIP_CHECK(hipMemcpy(h->descs_dev + (size_t)l * c.n_members + k, &ds[l], sizeof(SkinnyDesc), hipMemcpyHostToDevice));
    }
    if (h->enc_splitk) {
        SkinnyDesc sd{h->xpack, m.w_enc0, nullptr, nullptr, nullptr, nullptr, m.splitk, D, H, C, ND_ACT_NONE, 0};
        SplitKEpiDesc se{m.splitk, m.sc0, m.sh0, m.e0, H, 0 /* S is a launch argument */, ND_ACT_SOFTPLUS, opk};
        HIP_CHECK(hipMemcpy(h->spk_dev + k, &sd, sizeof sd, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(h->spke_dev + k, &se, sizeof se, hipMemcpyHostToDevice));
    }
    m.loaded = true;
    drop_graphs(h);
    return ND_OK;
}

static int check_range(nd_handle_s* h, int m0, int nm) {
    if (!h || !h->ws) return nd_set_err(ND_ERR_STATE, "workspace not bound");
    if (m0 < 0 || nm < 1 || m0 + nm > h->cfg.n_members) return nd_set_err(ND_ERR_ARG, "member range [%d,%d) invalid", m0, m0 + nm);
    for (int k = m0; k < m0 + nm; ++k)
        if (!h->members[k].loaded) return nd_set_err(ND_ERR_STATE, "member %d not loaded", k);
    return ND_OK;
}

template <int MODE>
static hipError_t launch_skinny(const SkinnyDesc* table, int K, int N, int M, int t, int nm, int half, hipStream_t st) {
    return nd_launch_skinny(nd_skinny_launch<MODE>(K, N, M, nm, half), SkinnyDesc{}, table, nm, M, t, st);
}

// One ConditionalLinear block (MODE 0: lin2, MODE 1: lin3 + lin4 projection) for nm members: the weight-streaming kernel for
// small row counts, the LDS-tiled kernel above 128 rows (nd_cond_gemm.hpp).  Returns the partial-sum count per (row, class)
// the MODE 1 form leaves in epart.
static int step_partials(const nd_handle_s* h, int M) {
    const int F = h->cfg.feature_dim;
    const CondGemmPlan p = nd_cond_gemm_plan(F, F, M, 1, h->half);
    return p.use_tile ? p.ntl : h->NT;
}

template <int MODE>
static hipError_t launch_step_block(nd_handle_s* h, const SkinnyDesc* table, int M, int t, int nm, hipStream_t st) {
    const int F = h->cfg.feature_dim;
    const CondGemmPlan p = nd_cond_gemm_plan(F, F, M, nm, h->half);
    if (p.use_tile) return nd_launch_cond_gemm(MODE, p, SkinnyDesc{}, table, M, t, h->tile_ws, st);
    return launch_skinny<MODE>(table, F, F, M, t, nm, h->half, st);
}

// "The inputs of this batch have been read": count the call on the device and publish the count to the caller's host-visible word
// (system-scope store: the word lives in pinned host memory).  Enqueued by nd_predict_batch right behind the LAST kernel that reads
// images_dev, so a loader may refill that buffer for the next batch while this batch's sampler is still running.
__global__ void k_inputs_consumed(unsigned* seq, unsigned* flag) {
    const unsigned v = *seq + 1u;
    *seq = v;
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int nd_set_input_flag(nd_handle h, uint32_t* flag_host_visible) {
    if (!h || !h->ws) return nd_set_err(ND_ERR_STATE, "workspace not bound");
    if ((uintptr_t)flag_host_visible & 3) return nd_set_err(ND_ERR_ARG, "flag must be 4-byte aligned");
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemset(h->input_seq, 0, sizeof(unsigned)));
    h->input_flag = flag_host_visible;
    drop_graphs(h);
    return ND_OK;
}

static int encode_impl(nd_handle h, int m0, int nm, const float* x_dev, int B, void* stream, bool signal_inputs);
extern "C" int nd_encode(nd_handle h, int m0, int nm, const float* x_dev, int B, void* stream) {
    return encode_impl(h, m0, nm, x_dev, B, stream, false);
}

static int encode_impl(nd_handle h, int m0, int nm, const float* x_dev, int B, void* stream, bool signal_inputs) {
    int rc = check_range(h, m0, nm);
    if (rc != ND_OK) return rc;
    if (!x_dev) return nd_set_err(ND_ERR_ARG, "x_dev is NULL");
    if (B < 1 || B > h->cfg.max_batch) return nd_set_err(ND_ERR_ARG, "B=%d outside [1,%d]", B, h->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    const nd_config& c = h->cfg;
    const int K = c.n_members, H = c.hidden_dim, F = c.feature_dim, D = c.data_dim;
    launch_pack(x_dev, h->xpack, B, D, h->half, st);       // images -> frag16 once; every member reads the same batch
    if (signal_inputs && h->input_flag)                    // nd_predict_batch: x_dev (= images_dev) is not read again by this call
        hipLaunchKernelGGL(k_inputs_consumed, dim3(1), dim3(1), 0, st, h->input_seq, h->input_flag);
    if (h->enc_splitk) {
        const SkinnyLaunch L = nd_skinny_launch<2>(D, H, B, nm, h->half);
        HIP_CHECK(nd_launch_skinny(L, SkinnyDesc{}, h->spk_dev + m0, nm, B, 0, st));
        const size_t q = (size_t)(((B + 15) / 16) * 16) * H / 4;
        hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((q + 255) / 256), 1, nm), dim3(256), 0, st, SplitKEpiDesc{},
                           (const SplitKEpiDesc*)(h->spke_dev + m0), B, L.S);
    } else {
        HIP_CHECK(launch_skinny<0>(h->descs_dev + (size_t)L_ENC0 * K + m0, D, H, B, 0, nm, h->half, st));
    }
    HIP_CHECK(launch_skinny<0>(h->descs_dev + (size_t)L_ENC1 * K + m0, H, H, B, 0, nm, h->half, st));
    HIP_CHECK(launch_skinny<0>(h->descs_dev + (size_t)L_ENC2 * K + m0, H, F, B, 0, nm, h->half, st));
    HIP_CHECK(hipGetLastError());
    h->encoded_B = B;
    return ND_OK;
}

extern "C" int nd_member_buffer(nd_handle h, int k, int which, float* dst_dev, int rows, void* stream) {
    if (!h || !h->ws || !dst_dev) return nd_set_err(ND_ERR_ARG, "bad argument");
    if (k < 0 || k >= h->cfg.n_members) return nd_set_err(ND_ERR_ARG, "member out of range");
    if (rows < 1 || rows > h->cfg.max_rows) return nd_set_err(ND_ERR_ARG, "rows out of range");
    MemberHost& m = h->members[k];
    const float* src = which == 0 ? m.xe : which == 1 ? m.h1 : which == 2 ? m.h2 : nullptr;
    if (!src) return nd_set_err(ND_ERR_ARG, "which=%d unknown", which);
    if (which == 0 && rows > h->cfg.max_batch) return nd_set_err(ND_ERR_ARG, "xe holds at most max_batch rows");
    const int F = h->cfg.feature_dim;
    // above 128 rows the step blocks of an fp32 handle run on frag32b3 images of h1 / h2 (bf16 matrix pipe): the live copies are
    // whichever the LAST launch over this member wrote (MemberHost::h_split), not what `rows` would pick
    if (which != 0 && m.h_split) return nd_join_rows(which == 1 ? m.h1s : m.h2s, dst_dev, rows, F, stream);
    const size_t n4 = (size_t)((rows + 15) / 16) * 16 * F / 4;
    if (h->half && which != 0)   // h1/h2 are GEMM operands (fp16 in that mode); xe never is
        hipLaunchKernelGGL(k_unpack_rows_h, dim3((unsigned)((n4 / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const _Float16*>(src), dst_dev, rows, F);
    else
        hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst_dev, rows, F);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

extern "C" long long nd_resident_weight_bytes(nd_handle h, int block) {
    if (!h || block < 0 || block > 1 || h->descs_host.empty()) return -1;
    const int K = h->cfg.n_members, F = h->cfg.feature_dim;
    long long n = 0;
    for (int k = 0; k < K; ++k)
        if (h->members[k].loaded && h->descs_host[(size_t)(block ? L_LIN3 : L_LIN2) * K + k].keep) n += (long long)F * F * (h->half ? 2 : 4);
    return n;
}

extern "C" int nd_set_profiling(nd_handle h, int enable) {
    if (!h) return nd_set_err(ND_ERR_ARG, "handle is NULL");
    if (h->profiling != (enable != 0)) drop_graphs(h);
    h->profiling = enable != 0;
    return ND_OK;
}

extern "C" int nd_set_loop_form(nd_handle h, int mode, float skew_us) {
    if (!h) return nd_set_err(ND_ERR_ARG, "handle is NULL");
    if (mode != 0 && mode != 1) return nd_set_err(ND_ERR_ARG, "mode must be 0 (per-step kernels) or 1 (one launch per loop)");
    if (skew_us > 1e6f) return nd_set_err(ND_ERR_ARG, "skew_us out of range");
    h->persist_mode = mode;
    if (skew_us >= 0.f) h->persist_skew_ticks = (int)(skew_us * 100.0f);
    drop_graphs(h);
    return ND_OK;
}

extern "C" int nd_loop_form(nd_handle h) { return h && h->persist_last ? 1 : 0; }

extern "C" int nd_persist_status(nd_handle h, int reset) {
    if (!h || !h->ws) return nd_set_err(ND_ERR_STATE, "workspace not bound");
    HIP_CHECK(hipDeviceSynchronize());
    unsigned flag = 0;
    HIP_CHECK(hipMemcpy(&flag, h->persist_bar + ND_PERSIST_ERR_WORD, sizeof flag, hipMemcpyDeviceToHost));
    if (reset && flag) HIP_CHECK(hipMemset(h->persist_bar, 0, ND_PERSIST_BAR_WORDS * sizeof(unsigned)));   // counters of the abandoned launch too
    return flag ? 1 : 0;
}

// event-record nodes of a graph about to be instantiated (-1 if the graph cannot be walked)
static int count_event_record_nodes(hipGraph_t g) {
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return -1;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return -1;
    int cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess) return -1;
        if (ty == hipGraphNodeTypeEventRecord) ++cnt;
    }
    return cnt;
}

extern "C" int nd_profile_probe_nodes(nd_handle h) {
    if (!h) return nd_set_err(ND_ERR_ARG, "handle is NULL");
    return h->probe_nodes;
}

extern "C" int nd_profile_read(nd_handle h, float* out_us, int* n_samples) {
    if (!h || !out_us || !n_samples) return nd_set_err(ND_ERR_ARG, "NULL argument");
    double acc[3] = {0, 0, 0};
    const int n = h->probe_steps;
    for (int s = 0; s < n; ++s)
        for (int k = 0; k < 3; ++k) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, h->probe_events[4 * s + k], h->probe_events[4 * s + k + 1]));
            acc[k] += ms * 1000.0;
        }
    out_us[0] = n ? (float)(acc[0] / n) : 0.f;      // head(i) interval              = head + o
    out_us[1] = n ? (float)(acc[1] / n) : 0.f;      // lin2 + lin3 (i) interval      = 2 blocks + o   (both launches, one record node)
    out_us[2] = n ? (float)(acc[2] / n) : 0.f;      // whole step i+1 interval       = head + 2 blocks + o
    const float head = out_us[2] - out_us[1];       // the head alone: the record overheads of the two intervals cancel
    out_us[3] = n ? (out_us[0] - head > 0.f ? out_us[0] - head : 0.f) : 0.f;     // o: what a record node adds to a loaded interval
    *n_samples = n;
    return ND_OK;
}

static int check_rows(nd_handle_s* h, int B, int mc, int T) {
    if (B < 1 || B > h->cfg.max_batch) return nd_set_err(ND_ERR_ARG, "B=%d outside [1,%d]", B, h->cfg.max_batch);
    if (mc < 1 || mc > 65535 || (long)B * mc > h->cfg.max_rows)
        return nd_set_err(ND_ERR_ARG, "mc=%d outside [1,65535] or B*mc=%ld exceeds max_rows=%d", mc, (long)B * mc, h->cfg.max_rows);
    if (T < 1 || T > h->cfg.n_steps) return nd_set_err(ND_ERR_ARG, "T=%d outside [1,%d]", T, h->cfg.n_steps);
    if (h->sched_T < T) return nd_set_err(ND_ERR_STATE, "schedule holds %d steps, need %d (nd_set_schedule)", h->sched_T, T);
    if (h->encoded_B != B) return nd_set_err(ND_ERR_STATE, "nd_encode ran with B=%d, sampling asks B=%d", h->encoded_B, B);
    return ND_OK;
}

// Which operand layout emit_loop picks for the step blocks of a launch over nm members at M rows (its `b9`): the frag32b3 images on the
// bf16 matrix pipe above 128 rows where the handle holds them and the descriptors travel by value.  Recorded per member at every entry
// point that runs the loop (graph replays included: the choice is a function of the call's shape) so that nd_member_buffer reads the
// copies the last launch really wrote.
static bool loop_writes_split(const nd_handle_s* h, int M, int nm) {
    return h->b9 && nm <= ND_INLINE_DESCS && nd_cond_gemm_plan(h->cfg.feature_dim, h->cfg.feature_dim, M, nm, h->half).use_tile;
}
static void note_h_layout(nd_handle_s* h, int m0, int nm, bool split) {
    for (int k = m0; k < m0 + nm; ++k) h->members[k].h_split = split;
}

// head -> lin2 -> lin3 for one member, then `final` in the requested mode (1 eps, 2 p_sample, 3 1to0)
static int single_eval(nd_handle_s* h, int member, StepIO io, int t, int final_mode, float* out, int B, int mc, hipStream_t st) {
    const nd_config& c = h->cfg;
    const int K = c.n_members, F = c.feature_dim, C = c.y_dim, M = B * mc;
    note_h_layout(h, member, 1, false);           // the per-member entry points always run the frag16 / frag32h forms into h1 / h2
    {
        const MemberDev* mdev = h->members_dev + member;
        MemberInline mi{};
        int mode = ND_HEAD_GIVEN, istep = 0, tprev = 0, tt = t, Bv = B, Mv = M, maxM = c.max_rows, Fv = F, NT = step_partials(h, M), Tn = c.n_steps;
        void* ah[] = {&mi, &mdev, &io, &mode, &istep, &tprev, &tt, &Bv, &Mv, &maxM, &Fv, &NT, &Tn};
        HIP_CHECK(hipLaunchKernel(head_fn(C), dim3((F + 1023) / 1024, M, 1), dim3(256), ah, 0, st));
    }
    HIP_CHECK(launch_step_block<0>(h, h->descs_dev + (size_t)L_LIN2 * K + member, M, t, 1, st));
    HIP_CHECK(launch_step_block<1>(h, h->descs_dev + (size_t)L_LIN3 * K + member, M, t, 1, st));
    {
        const MemberDev* mdev = h->members_dev + member;
        MemberInline mi{};
        int fm = final_mode, tt = t, Bv = B, Mv = M, maxM = c.max_rows, NT = step_partials(h, M), Tn = c.n_steps;
        size_t ems = 0;
        void* af[] = {&mi, &mdev, &io, &fm, &tt, &Bv, &Mv, &maxM, &NT, &Tn, &out, &ems};
        HIP_CHECK(hipLaunchKernel(final_fn(C), dim3(M, 1, 1), dim3(64), af, 0, st));
    }
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

extern "C" int nd_eps_theta(nd_handle h, int member, const float* y_dev, const float* yhat_dev, int t, float* eps_out, int B,
                            int mc, void* stream) {
    int rc = check_range(h, member, 1);
    if (rc != ND_OK) return rc;
    if (!y_dev || !yhat_dev || !eps_out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (B < 1 || B > h->cfg.max_batch || mc < 1 || (long)B * mc > h->cfg.max_rows) return nd_set_err(ND_ERR_ARG, "B/mc out of range");
    if (t < 0 || t >= h->cfg.n_steps) return nd_set_err(ND_ERR_ARG, "t=%d outside [0,%d)", t, h->cfg.n_steps);
    if (h->encoded_B != B) return nd_set_err(ND_ERR_STATE, "nd_encode ran with B=%d, asked B=%d", h->encoded_B, B);
    StepIO io{};
    io.yhat = yhat_dev; io.y_in = y_dev; io.alphas = h->alphas; io.omabs = h->omabs;
    return single_eval(h, member, io, t, 1, eps_out, B, mc, (hipStream_t)stream);
}

extern "C" int nd_p_sample(nd_handle h, int member, const float* y_dev, const float* yhat_dev, const float* ymean_dev,
                           const float* z_dev, int t, float* y_out, int B, int mc, void* stream) {
    int rc = check_range(h, member, 1);
    if (rc != ND_OK) return rc;
    if (!y_dev || !yhat_dev || !ymean_dev || !y_out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (t < 0) return nd_set_err(ND_ERR_ARG, "t must be >= 0");
    if (t > 0 && !z_dev) return nd_set_err(ND_ERR_ARG, "z_dev is required for t >= 1");
    rc = check_rows(h, B, mc, t + 1);
    if (rc != ND_OK) return rc;
    StepIO io{};
    io.yhat = yhat_dev; io.ymean = ymean_dev; io.y_in = y_dev; io.noise = z_dev; io.alphas = h->alphas; io.omabs = h->omabs;
    return single_eval(h, member, io, t, t > 0 ? 2 : 3, y_out, B, mc, (hipStream_t)stream);
}

// Enqueue (eager) or record (graph) the 3T+1 kernels of one p_sample_loop for a member range.
struct Emitter {
    hipStream_t st;
    hipGraph_t graph = nullptr;
    hipGraphNode_t last = nullptr;
    hipError_t err = hipSuccess;
    bool capturing = false;          // st is being captured into a graph (nd_predict_batch)
    void record(hipEvent_t ev) {
        if (err != hipSuccess) return;
        if (!graph && capturing) {
            // Under stream capture a plain hipEventRecord only orders the captured work: nothing would stamp the event when the
            // graph is replayed.  Put an event-record NODE into the graph being captured, behind the stream's current dependency
            // set, and make it the stream's new dependency set (hipEventRecordWithFlags(..., hipEventRecordExternal) is the short
            // form of this, but returns hipErrorInvalidValue under relaxed-mode capture on ROCm 7.2).
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            unsigned long long id = 0;
            hipGraph_t g = nullptr;
            const hipGraphNode_t* deps = nullptr;
            size_t ndeps = 0;
            err = hipStreamGetCaptureInfo_v2(st, &cs, &id, &g, &deps, &ndeps);
            if (err != hipSuccess) return;
            if (cs != hipStreamCaptureStatusActive || !g) { err = hipErrorStreamCaptureInvalidated; return; }
            hipGraphNode_t node;
            err = hipGraphAddEventRecordNode(&node, g, deps, ndeps, ev);
            if (err != hipSuccess) return;
            err = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
        } else if (!graph) {
            err = hipEventRecord(ev, st);
        } else {
            hipGraphNode_t node;
            err = hipGraphAddEventRecordNode(&node, graph, last ? &last : nullptr, last ? 1 : 0, ev);
            last = node;
        }
    }
    void emit(void* fn, dim3 grid, dim3 block, void** args, size_t lds = 0) {
        if (err != hipSuccess) return;
        if (!graph) {
            err = hipLaunchKernel(fn, grid, block, args, lds, st);
        } else {
            hipKernelNodeParams p{};
            p.func = fn; p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = (unsigned)lds; p.kernelParams = args; p.extra = nullptr;
            hipGraphNode_t node;
            err = hipGraphAddKernelNode(&node, graph, last ? &last : nullptr, last ? 1 : 0, &p);
            last = node;
        }
    }
};

static hipError_t emit_loop(nd_handle_s* h, Emitter& em, int m0, int nm, StepIO io, int B, int mc, int T) {
    const nd_config& c = h->cfg;
    int K = c.n_members, F = c.feature_dim, C = c.y_dim, M = B * mc, maxM = c.max_rows, NT = step_partials(h, M), Tn = T;
    // descriptor rows by value while they fit (nm <= ND_INLINE_DESCS), else through the device tables
    const bool inl = nm <= ND_INLINE_DESCS;
    const MemberDev* mdev = inl ? nullptr : h->members_dev + m0;
    const SkinnyDesc* t2 = h->descs_dev + (size_t)L_LIN2 * K + m0;
    const SkinnyDesc* t3 = h->descs_dev + (size_t)L_LIN3 * K + m0;
    const SkinnyDesc *s2 = inl ? nullptr : t2, *s3 = inl ? nullptr : t3;     // k_skinny's table argument
    MemberInline mi{};
    SkinnyInline i2{}, i3{};
    for (int g = 0; inl && g < nm; ++g) {
        mi.m[g] = h->members_host[m0 + g];
        i2.d[g] = h->descs_host[(size_t)L_LIN2 * K + m0 + g];
        i3.d[g] = h->descs_host[(size_t)L_LIN3 * K + m0 + g];
    }
    // ONE launch for the whole loop where the plan allows (csrc/nd_persist.hip)
    h->persist_last = false;
    if (h->persist_mode && inl) {
        const PersistPlan pp = nd_persist_plan(F, M, nm, C, h->half);
        if (pp.ok) {
            if (pp.err != hipSuccess) return pp.err;
            PersistArgs pa{};
            for (int g = 0; g < nm; ++g) { pa.l2[g] = i2.d[g]; pa.l3[g] = i3.d[g]; pa.mem[g] = mi.m[g]; }
            pa.io = io;
            const char* act = getenv("ND_PERSIST_ACTIVE");                  // experiments only: run the first n members of the launch
            const char* spin = getenv("ND_PERSIST_SPIN_TICKS");             // tests only: how long a barrier wait may last (100 MHz ticks)
            const char* fake = getenv("ND_PERSIST_FAKE_RESIDENT");          // timing ablation only: results are wrong (nd_persist.hpp)
            pa.s = PersistScalars{h->persist_bar, nm, B, M, maxM, F, T, h->persist_skew_ticks, spin ? atoi(spin) : 100000000 /* 1 s */,
                                  act ? atoi(act) : nm, fake ? atoi(fake) : 0};
            note_h_layout(h, m0, nm, false);
            hipEvent_t* ev = nullptr;
            if (h->profiling) {
                if (h->probe_events.size() < 4) {
                    const size_t old = h->probe_events.size();
                    h->probe_events.resize(4);
                    for (size_t e = old; e < 4; ++e)
                        if (hipEventCreate(&h->probe_events[e]) != hipSuccess) return hipErrorOutOfMemory;
                }
                ev = h->probe_events.data();
            }
            if (ev) em.record(ev[0]);
            void* ap[] = {&pa};
            em.emit(pp.fn, pp.grid, pp.block, ap, pp.lds);
            if (ev) em.record(ev[1]);
            h->probe_steps = 0;
            h->persist_last = true;
            return em.err;
        }
    }
    SkinnyDesc d0{};
    const dim3 ghead((F + 1023) / 1024, M, nm);
    const SkinnyLaunch L2 = nd_skinny_launch<0>(F, F, M, nm, h->half), L3 = nd_skinny_launch<1>(F, F, M, nm, h->half);
    if (L2.err != hipSuccess) return L2.err;       // the kernel's dynamic-LDS attribute could not be set on this device
    if (L3.err != hipSuccess) return L3.err;
    int cps2 = L2.cps, cps3 = L3.cps;
    // more than 128 rows: the LDS-tiled kernel (+ its fixup for the k-split tail) takes the place of each k_skinny node
    CondGemmPlan tp = nd_cond_gemm_plan(F, F, M, nm, h->half);
    float* tws = h->tile_ws;
    // ... on the bf16 matrix pipe (frag32b3 operands, exact fp32 products) where the handle holds the images: the step head then
    // writes h1 as such an image (its record travels by value, so the layout is chosen per launch; the device-table path of more
    // than ND_INLINE_DESCS members keeps the f32-input MFMA kernel)
    const bool b9 = tp.use_tile && h->b9 && inl;
    note_h_layout(h, m0, nm, b9);
    if (b9) {
        t2 = h->descs_dev + (size_t)L_LIN2S * K + m0;
        t3 = h->descs_dev + (size_t)L_LIN3S * K + m0;
        for (int g = 0; g < nm; ++g) { mi.m[g].h1 = (float*)h->members[m0 + g].h1s; mi.m[g].h16 = 2; }
    }
    const dim3 tgrid((unsigned)(tp.n_full + tp.rem * tp.split)), tfix((unsigned)tp.rem * (b9 ? 16 : 4));
    // ... and the head in its many-rows form (16 rows per workgroup, whole operand planes per store; same bits)
    const bool rows_head = b9 && NT <= 64 && !getenv("ND_HEAD_PER_ROW");
    const dim3 ghead_rows((F + ND_HEAD_ROWS_COLS - 1) / ND_HEAD_ROWS_COLS, (M + 15) / 16, nm);
    // probes: up to 8 PAIRS of steps (i, i+1) spread over the loop (never step 0: its head is the cheap INIT form).  Records
    // around head(i), around the two blocks of step i, and behind the blocks of step i+1: the third interval is one whole unrecorded
    // step, so  (whole step) - (two blocks) = the head alone, and what a record node adds to an interval follows from the head's
    // own interval -- calibrated inside the loaded graph instead of by an empty interval (two record nodes back to back cost 6 us,
    // most of which a kernel launched behind a record node hides).
    const int want = h->profiling ? ((T - 1) / 2 < 8 ? (T - 1) / 2 : 8) : 0;
    const int stride = want > 0 ? (T - 1) / want : 0;
    hipEvent_t pending_end = nullptr;
    if ((int)h->probe_events.size() < 4 * want) {
        const size_t old = h->probe_events.size();
        h->probe_events.resize(4 * want);
        for (size_t e = old; e < h->probe_events.size(); ++e)
            if (hipEventCreate(&h->probe_events[e]) != hipSuccess) return hipErrorOutOfMemory;
    }
    int probed = 0;
    for (int i = 0; i < T; ++i) {
        int t = T - 1 - i, t_prev = t + 1, mode = (i == 0) ? ND_HEAD_INIT : ND_HEAD_UPDATE, istep = i;
        const bool probe = want > 0 && i >= 1 && i + 1 < T && !pending_end && probed < want && ((i - 1) % stride) == (stride - 1) / 2;
        hipEvent_t* ev = probe ? &h->probe_events[4 * probed] : nullptr;
        hipEvent_t end_of_pair = pending_end;      // this step closes the pair opened by the previous one
        pending_end = nullptr;
        if (probe) em.record(ev[0]);
        void* ah[] = {&mi, &mdev, &io, &mode, &istep, &t_prev, &t, &B, &M, &maxM, &F, &NT, &Tn};
        em.emit(rows_head ? head_rows_fn(C) : head_fn(C), rows_head ? ghead_rows : ghead, dim3(256), ah);
        if (probe) em.record(ev[1]);
        if (tp.use_tile) {
            void* a2[] = {&d0, &t2, &M, &t, &tp.TM, &tp.TN, &tp.n_full, &tp.split, &tws};
            void* a3[] = {&d0, &t3, &M, &t, &tp.TM, &tp.TN, &tp.n_full, &tp.split, &tws};
            if (b9) {
                em.emit(nd_cond_gemm_b9_kernel(0), tgrid, dim3(512), a2, nd_cond_gemm_b9_dynlds());
                if (tp.rem > 0) em.emit(nd_cond_gemm_b9_fixup_kernel(0), tfix, dim3(64), a2);
                em.emit(nd_cond_gemm_b9_kernel(1), tgrid, dim3(512), a3, nd_cond_gemm_b9_dynlds());
                if (tp.rem > 0) em.emit(nd_cond_gemm_b9_fixup_kernel(1), tfix, dim3(64), a3);
            } else {
                em.emit(nd_cond_gemm_kernel(0), tgrid, dim3(256), a2, nd_cond_gemm_dynlds());
                if (tp.rem > 0) em.emit(nd_cond_gemm_fixup_kernel(0), tfix, dim3(64), a2);
                em.emit(nd_cond_gemm_kernel(1), tgrid, dim3(256), a3, nd_cond_gemm_dynlds());
                if (tp.rem > 0) em.emit(nd_cond_gemm_fixup_kernel(1), tfix, dim3(64), a3);
            }
        } else {
            void* a2[] = {&i2, &s2, &nm, &M, &t, &cps2};
            em.emit(L2.fn, L2.grid, L2.block, a2, L2.lds);
            void* a3[] = {&i3, &s3, &nm, &M, &t, &cps3};
            em.emit(L3.fn, L3.grid, L3.block, a3, L3.lds);
        }
        // the two ConditionalLinear launches share ONE interval (one record node for two kernels: half the distortion per launch)
        if (probe) { em.record(ev[2]); pending_end = ev[3]; ++probed; }
        if (end_of_pair) em.record(end_of_pair);
    }
    h->probe_steps = probed;
    int eps_only = 0, par_cur = (T - 1) & 1;
    float* eps_out = nullptr;
    size_t eps_ms = 0;
    void* af[] = {&mi, &mdev, &io, &eps_only, &par_cur, &B, &M, &maxM, &NT, &Tn, &eps_out, &eps_ms};
    em.emit(final_fn(C), dim3(M, 1, nm), dim3(64), af);
    return em.err;
}

// Enqueue (eager, or under stream capture) the kernels of one sampling call: [Philox fill] -> 3T+1 step kernels -> [advance].
static int run_loop_eager(nd_handle_s* h, hipStream_t st, int m0, int nm, StepIO io, bool draw, int B, int mc, int T) {
    if (draw) HIP_CHECK(nd_launch_philox_normal(h->noise_ws, h->rng_state, 0, 0, 0, m0, nm, T, B, mc, h->cfg.y_dim, st));
    Emitter em{st};
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    HIP_CHECK(hipStreamIsCapturing(st, &cs));
    em.capturing = cs == hipStreamCaptureStatusActive;
    hipError_t e = emit_loop(h, em, m0, nm, io, B, mc, T);
    if (e != hipSuccess) return nd_set_err(ND_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    if (draw) HIP_CHECK(nd_launch_rng_advance(h->rng_state, st));
    return ND_OK;
}

extern "C" int nd_sample(nd_handle h, int m0, int nm, const float* yhat_dev, const float* ymean_dev, const float* noise_dev,
                         float* y0_out_dev, float* seq_out_dev, int B, int mc, int T, int use_graph, void* stream) {
    int rc = check_range(h, m0, nm);
    if (rc != ND_OK) return rc;
    if (!yhat_dev || !ymean_dev || !y0_out_dev) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    rc = check_rows(h, B, mc, T);
    if (rc != ND_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int C = h->cfg.y_dim, M = B * mc;
    // noise_dev == NULL: the draws come from the library's counter-based generator (nd_seed), written to the workspace by the
    // first node and consumed from there; the last node advances the batch counter
    const bool draw = noise_dev == nullptr;
    StepIO io{};
    io.yhat = yhat_dev;   io.yhat_ms = (size_t)B * C;
    io.ymean = ymean_dev; io.ymean_ms = (size_t)B * C;
    io.noise = draw ? h->noise_ws : noise_dev; io.noise_ms = (size_t)T * M * C;
    io.y0_out = y0_out_dev; io.y0_ms = (size_t)M * C;
    io.seq_out = seq_out_dev; io.seq_ms = (size_t)(T + 1) * M * C;
    io.alphas = h->alphas; io.omabs = h->omabs;
    if (!use_graph) return run_loop_eager(h, st, m0, nm, io, draw, B, mc, T);
    GraphKey key{m0, nm, B, mc, T, yhat_dev, ymean_dev, noise_dev, y0_out_dev, seq_out_dev};
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        Emitter em{st};
        HIP_CHECK(hipGraphCreate(&em.graph, 0));
        if (draw) {
            float* out = h->noise_ws;
            const unsigned long long* state = h->rng_state;
            unsigned long long seed0 = 0;
            uint32_t z0 = 0, z1 = 0;
            int m0v = m0, nmv = nm, Tv = T, Bv = B, mcv = mc, Cv = C;
            void* a[] = {&out, &state, &seed0, &z0, &z1, &m0v, &nmv, &Tv, &Bv, &mcv, &Cv};
            const size_t total = (size_t)nm * T * M * ((C + 3) / 4);
            em.emit(nd_philox_normal_kernel(), dim3((unsigned)((total + 255) / 256)), dim3(256), a);
        }
        hipError_t e = emit_loop(h, em, m0, nm, io, B, mc, T);
        if (e == hipSuccess && draw) {
            unsigned long long* state = h->rng_state;
            void* a[] = {&state};
            em.emit(nd_rng_advance_kernel(), dim3(1), dim3(64), a);
            e = em.err;
        }
        if (e != hipSuccess) {
            (void)hipGraphDestroy(em.graph);
            return nd_set_err(ND_ERR_HIP, "graph build failed: %s", hipGetErrorString(e));
        }
        h->probe_nodes = count_event_record_nodes(em.graph);
        hipGraphExec_t exec;
        e = hipGraphInstantiate(&exec, em.graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(em.graph);
        if (e != hipSuccess) return nd_set_err(ND_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
        if (h->graphs.size() >= 64) drop_graphs(h);
        it = h->graphs.emplace(key, exec).first;
    }
    HIP_CHECK(hipGraphLaunch(it->second, st));
    note_h_layout(h, m0, nm, loop_writes_split(h, M, nm));
    return ND_OK;
}

// ---------------------------------------------------------------------------------------------
// The whole hot path of one test batch (classification_train_separately.py:749-794) as one call / one hipGraph:
//   :753      compute_guiding_prediction           nd_guiding_prediction (ViT prefix + mapping MLPs + softmax, :755-758)
//   :747      images_224_flat                      the same buffer viewed [B, D]
//   (hoist)   xe = norm(encoder_x(x)) per member   nd_encode, once per (member, batch) instead of once per step
//   :767-777  K members x mc trials p_sample_loop  the 3T+1 step kernels (all members and trials per launch)
//   :786-789  vote, compute_ensemble_confidence    nd_aggregate
// The graph is recorded by stream capture on a private stream (the caller's stream may be the legacy default stream, which
// cannot be captured) after one eager pass, which also makes every lazily set function attribute (dynamic LDS sizes) exist
// before the capture; it is then replayed on the caller's stream.
// ---------------------------------------------------------------------------------------------
static int batch_enqueue(nd_handle_s* h, nd_cond c, const float* images, const float* noise, const nd_batch_out* out, int B, int mc,
                         int T, float temperature, hipStream_t st) {
    const int K = h->cfg.n_members, C = h->cfg.y_dim, M = B * mc;
    const bool draw = noise == nullptr;
    int rc = nd_guiding_prediction_first(c, images, h->logits_ws, out->yhat, B, K, st);     // member k <- mapping MLP k, k < K
    if (rc != ND_OK) return rc;
    rc = encode_impl(h, 0, K, images, B, st, true);       // (the conditioner's im2col ran before: the pack here is the last reader of `images`)
    if (rc != ND_OK) return rc;
    StepIO io{};
    io.yhat = out->yhat;  io.yhat_ms = (size_t)B * C;
    io.ymean = out->yhat; io.ymean_ms = (size_t)B * C;           // y_T_mean = y_0_hat (:762, quirk Q2)
    io.noise = draw ? h->noise_ws : noise; io.noise_ms = (size_t)T * M * C;
    io.y0_out = out->samples; io.y0_ms = (size_t)M * C;          // [K][mc*B][C] == [K*mc][B][C]: member-major, then trial
    io.alphas = h->alphas; io.omabs = h->omabs;
    rc = run_loop_eager(h, st, 0, K, io, draw, B, mc, T);
    if (rc != ND_OK) return rc;
    return nd_aggregate(out->samples, out->prob, out->vote, out->probs, K * mc, B, C, temperature, st);
}

extern "C" int nd_predict_batch(nd_handle h, nd_cond c, const float* images_dev, const float* noise_dev, const nd_batch_out* out,
                                int B, int mc, int T, float temperature, int use_graph, void* stream) {
    if (!h || !h->ws) return nd_set_err(ND_ERR_STATE, "workspace not bound");
    if (!c) return nd_set_err(ND_ERR_ARG, "conditioner is NULL");
    if (!images_dev || !out || !out->samples || !out->prob || !out->vote || !out->yhat) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    const nd_cond_config* cc = nd_cond_get_config(c);
    const nd_config& g = h->cfg;
    if (cc->n_mlps < g.n_members) return nd_set_err(ND_ERR_ARG, "conditioner has %d mapping MLPs, ensemble %d members", cc->n_mlps, g.n_members);
    if (cc->num_classes != g.y_dim) return nd_set_err(ND_ERR_ARG, "conditioner has %d classes, ensemble %d", cc->num_classes, g.y_dim);
    if ((long)cc->in_chans * cc->img_size * cc->img_size != g.data_dim)
        return nd_set_err(ND_ERR_ARG, "image size %dx%dx%d != data_dim %d", cc->in_chans, cc->img_size, cc->img_size, g.data_dim);
    int rc = check_range(h, 0, g.n_members);
    if (rc != ND_OK) return rc;
    if (B < 1 || B > g.max_batch || B > cc->max_batch) return nd_set_err(ND_ERR_ARG, "B=%d outside [1,%d]", B, g.max_batch < cc->max_batch ? g.max_batch : cc->max_batch);
    if (mc < 1 || mc > 65535 || (long)B * mc > g.max_rows)
        return nd_set_err(ND_ERR_ARG, "mc=%d outside [1,65535] or B*mc=%ld exceeds max_rows=%d", mc, (long)B * mc, g.max_rows);
    if (T < 1 || T > g.n_steps) return nd_set_err(ND_ERR_ARG, "T=%d outside [1,%d]", T, g.n_steps);
    if (h->sched_T < T) return nd_set_err(ND_ERR_STATE, "schedule holds %d steps, need %d (nd_set_schedule)", h->sched_T, T);
    if (!(temperature > 0.f)) return nd_set_err(ND_ERR_ARG, "temperature must be > 0");
    hipStream_t st = (hipStream_t)stream;
    if (!use_graph) return batch_enqueue(h, c, images_dev, noise_dev, out, B, mc, T, temperature, st);
    unsigned tb;
    memcpy(&tb, &temperature, sizeof tb);
    BatchKey key{nd_cond_serial(c), images_dev, noise_dev, out->samples, out->prob, out->vote, out->probs, out->yhat, B, mc, T, tb, h->profiling};
    auto it = h->batch_graphs.find(key);
    if (it == h->batch_graphs.end()) {
        // first call of this shape: one eager pass on the caller's stream IS this call's result; the graph recorded next to it is
        // for the following calls
        rc = batch_enqueue(h, c, images_dev, noise_dev, out, B, mc, T, temperature, st);
        if (rc != ND_OK) return rc;
        if (!h->capture_stream) HIP_CHECK(hipStreamCreateWithFlags(&h->capture_stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeRelaxed));
        rc = batch_enqueue(h, c, images_dev, noise_dev, out, B, mc, T, temperature, h->capture_stream);
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(h->capture_stream, &graph);
        if (rc != ND_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e != hipSuccess) return nd_set_err(ND_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        h->probe_nodes = count_event_record_nodes(graph);
        hipGraphExec_t exec;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return nd_set_err(ND_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
        if (h->batch_graphs.size() >= 16) drop_graphs(h);
        h->batch_graphs.emplace(key, exec);
        h->encoded_B = B;
        return ND_OK;
    }
    HIP_CHECK(hipGraphLaunch(it->second, st));
    note_h_layout(h, 0, g.n_members, loop_writes_split(h, B * mc, g.n_members));
    h->encoded_B = B;
    return ND_OK;
}
