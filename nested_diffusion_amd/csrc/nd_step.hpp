// nd_step.hpp -- what the kernels of one denoising step share: the per-member device record, the per-launch tensor table and the
// pinned arithmetic of the posterior update and of the first ConditionalLinear block (csrc/nd_sampler.hip: the per-step kernels of the
// hipGraph form; csrc/nd_persist.hip: the one-launch form of the same loop).  gfx950 only.
//
// Reference (file:line relative to the reference checkout): diffusion/diffusion_utils.py:54-111, diffusion/latent_model.py:173-177.
#pragma once
#include "nd_common.hpp"

// ---------------------------------------------------------------------------------------------
// device-side member record (everything the head / final kernels need)
// ---------------------------------------------------------------------------------------------
struct MemberDev {
    const float* lin1_w;   // [F, 2C] (workspace copy)
    const float* lin4_b;   // [C]
    const float* A1;       // [T, F] folded gain  (unetnorm1 scale * embed1[t])
    const float* C1;       // [T, F] folded shift
    const float* xe;       // frag16 [B, F]
    float* h1;             // frag16 [M, F]
    float* ybuf;           // [2, maxM, C]
    const float* epart;    // [M, C, NT]
    int h16;               // layout h1 is written in: 0 frag16 fp32, 1 frag32h fp16, 2 frag32b3 (three bf16 pieces per value, csrc/nd_b9.hpp:
                           // the input of the lin2 block on the bf16 matrix pipe; h1 then points at that image)
};

#define ND_MAX_C 8
struct MemberInline { MemberDev m[ND_INLINE_DESCS]; };   // by value in the kernel arguments (members == nullptr): see SkinnyInline

// Pointers that come out of descriptor structs are generic to the compiler; loads through them become flat_load, which is
// counted on vmcnt AND lgkmcnt and cannot be waited on selectively -- the step head's "tables in flight under the reduction"
// would serialise at the first LDS access.  These casts put the accesses in the global address space.
typedef const __attribute__((address_space(1))) float* nd_gcf;
typedef __attribute__((address_space(1))) float* nd_gf;
#define ND_GC(p) ((nd_gcf)(p))
#define ND_GW(p) ((nd_gf)(p))

struct StepIO {             // per-launch tensors with a member-major leading stride
    const float* yhat;  size_t yhat_ms;    // [nm][B][C]
    const float* ymean; size_t ymean_ms;   // [nm][B][C]
    const float* noise; size_t noise_ms;   // [nm][T][M][C]
    float* y0_out;      size_t y0_ms;      // [nm][M][C]
    float* seq_out;     size_t seq_ms;     // [nm][T+1][M][C] or null
    const float* y_in;  size_t yin_ms;     // [nm][M][C] (eps_theta entry point only)
    const float* alphas; const float* omabs;
};

// diffusion_utils.py:68-92 in the reference's operation order, fp32, no FMA contraction, so the
// posterior is bit-identical to the CPU path for identical eps.
__device__ __forceinline__ float nd_posterior(float y, float ymean, float eps, float z, float alpha_t, float s_t,
                                              float s_tm1) {
#pragma clang fp contract(off)
    const float st2 = s_t * s_t;
    const float sab_t = sqrtf(1.0f - st2);
    const float stm2 = s_tm1 * s_tm1;
    const float sab_tm1 = sqrtf(1.0f - stm2);
    const float sa = sqrtf(alpha_t);
    const float g0 = (1.0f - alpha_t) * sab_tm1 / st2;
    const float g1 = stm2 * sa / st2;
    const float g2 = 1.0f + (sab_t - 1.0f) * (sa + sab_tm1) / st2;
    const float y0r = 1.0f / sab_t * (y - (1.0f - sab_t) * ymean - eps * s_t);
    const float mean = g0 * y0r + g1 * y + g2 * ymean;
    const float bh = stm2 / st2 * (1.0f - alpha_t);
    return mean + sqrtf(bh) * z;
}

// diffusion_utils.py:99-111
__device__ __forceinline__ float nd_y0_reparam(float y, float ymean, float eps, float s_t) {
#pragma clang fp contract(off)
    const float sab_t = sqrtf(1.0f - s_t * s_t);
    return 1.0f / sab_t * (y - (1.0f - sab_t) * ymean - eps * s_t);
}

// One element of h1 = softplus(A1[t] * (lin1.W [y_t, yhat]) + C1[t]) * xe (latent_model.py:173-177 after the folds of SURVEY 7.3):
// a pinned sequence of FMAs, so the two step-head kernels return the same bits.  w: the 2C entries of lin1.weight's row.
template <int C, typename WT>
__device__ __forceinline__ float nd_head_element(const WT& w, const float (&yv)[C], const float (&yh)[C], float a, float cc, float xe) {
#pragma clang fp contract(off)
    float u = 0.f;
#pragma unroll
    for (int q = 0; q < C; ++q) u = __builtin_fmaf(w[q], yv[q], u);
#pragma unroll
    for (int q = 0; q < C; ++q) u = __builtin_fmaf(w[C + q], yh[q], u);
    return nd_softplus(__builtin_fmaf(a, u, cc)) * xe;
}

