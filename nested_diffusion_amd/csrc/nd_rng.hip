// nd_rng.hip -- in-library Gaussian noise for the sampler's throughput mode (gfx950 only).
//
// Reference: the draws of p_sample_loop / p_sample (diffusion/diffusion_utils.py:139 `z = torch.randn_like(y_T_mean)`, :67
// `z = torch.randn_like(y)`), T draws of [B, C] per (member, trial).  torch's CPU generator (mt19937) cannot be reproduced on a
// device, so parity runs hand the draws in; this file is what replaces them when the caller passes no noise tensor.
//
// Generator: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), counter-based:
//   key     = (seed low word, seed high word)
//   counter = (global image index, trial | member << 16 | class-quad << 24, draw index i, batch counter)
// so the noise of one (image, member, trial, step) is a pure function of its indices: it does not depend on how a batch is split
// over ranks or on launch geometry.  The four 32-bit outputs become four standard normals by Box-Muller:
//   u1 = (x0 + 1) * 2^-32 in (0, 1], u2 = x1 * 2^-32 in [0, 1):  r = sqrt(-2 ln u1), (z0, z1) = r * (cos, sin)(2 pi u2); same for (x2, x3).
#include "nd_rng.hpp"
#include "../../include/nested_diffusion.h"

int nd_set_err(int code, const char* fmt, ...);
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return nd_set_err(ND_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

__device__ __forceinline__ void nd_philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void nd_box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;       // (0, 1]: (2^32 - 1) + 1 rounds to 2^32 -> exactly 1
    const float u2 = (float)b * 2.3283064365386963e-10f;                 // [0, 1]
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincospif(2.0f * u2, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

// One thread per (member, draw, row, class-quad).  state: {seed, batch counter | first image << 32} read from the device when
// `state` is non-null (so a replayed hipGraph sees the current values), else taken from the arguments.
__global__ __launch_bounds__(256) void k_philox_normal(float* __restrict__ out, const unsigned long long* __restrict__ state,
                                                       unsigned long long seed_arg, uint32_t batch_arg, uint32_t first_arg,
                                                       int m0, int nm, int T, int B, int mc, int C) {
    const int Q = (C + 3) / 4, M = B * mc;
    const size_t total = (size_t)nm * T * M * Q;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    unsigned long long seed = seed_arg;
    uint32_t batch = batch_arg, first = first_arg;
    if (state) {
        seed = state[0];
        batch = (uint32_t)state[1];
        first = (uint32_t)(state[1] >> 32);
    }
    const int q = (int)(idx % Q);
    const int m = (int)((idx / Q) % M);
    const int i = (int)((idx / ((size_t)Q * M)) % T);
    const int k = (int)(idx / ((size_t)Q * M * T));
    const int trial = m / B, b = m % B;
    // member index = its position in the ENSEMBLE (m0 + k), not in this call's range: a member's draws do not depend on how the
    // members are split over calls
    uint32_t c[4] = {first + (uint32_t)b, (uint32_t)trial | ((uint32_t)(m0 + k) << 16) | ((uint32_t)q << 24), (uint32_t)i, batch};
    nd_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float z[4];
    nd_box_muller(c[0], c[1], z[0], z[1]);
    nd_box_muller(c[2], c[3], z[2], z[3]);
    float* o = out + (((size_t)k * T + i) * M + m) * C + 4 * q;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * q + j < C) o[j] = z[j];
}

__global__ void k_rng_advance(unsigned long long* state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[1] = (state[1] & 0xFFFFFFFF00000000ull) | (uint32_t)((uint32_t)state[1] + 1u);
}

__global__ void k_philox_raw(const uint32_t* __restrict__ ctr, uint32_t* __restrict__ out, int n, uint32_t k0, uint32_t k1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c[4] = {ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3]};
    nd_philox4x32_10(c, k0, k1);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[4 * i + j] = c[j];
}

hipError_t nd_launch_philox_normal(float* out, const unsigned long long* state_dev, unsigned long long seed, uint32_t batch, uint32_t first,
                                   int m0, int nm, int T, int B, int mc, int C, hipStream_t st) {
    const size_t total = (size_t)nm * T * B * mc * ((C + 3) / 4);
    hipLaunchKernelGGL(k_philox_normal, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, state_dev, seed, batch, first, m0, nm, T, B, mc, C);
    return hipGetLastError();
}

hipError_t nd_launch_rng_advance(unsigned long long* state_dev, hipStream_t st) {
    hipLaunchKernelGGL(k_rng_advance, dim3(1), dim3(64), 0, st, state_dev);
    return hipGetLastError();
}

void* nd_philox_normal_kernel() { return (void*)k_philox_normal; }
void* nd_rng_advance_kernel() { return (void*)k_rng_advance; }

extern "C" int nd_philox_normal(float* out_dev, int n_members, int T, int B, int mc, int C, uint64_t seed, uint32_t batch_counter,
                                uint32_t first_image, void* stream) {
    if (!out_dev) return nd_set_err(ND_ERR_ARG, "out_dev is NULL");
    // counter word 1 = trial (16 bits) | member (8 bits) | class-quad (8 bits)
    if (n_members < 1 || n_members > 255 || T < 1 || B < 1 || mc < 1 || mc > 65535 || C < 1 || C > 1024)
        return nd_set_err(ND_ERR_ARG, "need 1 <= n_members <= 255, 1 <= mc <= 65535, T, B >= 1, 1 <= C <= 1024");
    HIP_CHECK(nd_launch_philox_normal(out_dev, nullptr, seed, batch_counter, first_image, 0, n_members, T, B, mc, C, (hipStream_t)stream));
    return ND_OK;
}

extern "C" int nd_philox_raw(const uint32_t* ctr_dev, uint32_t* out_dev, int n, uint32_t key0, uint32_t key1, void* stream) {
    if (!ctr_dev || !out_dev || n < 1) return nd_set_err(ND_ERR_ARG, "bad philox_raw arguments");
    hipLaunchKernelGGL(k_philox_raw, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctr_dev, out_dev, n, key0, key1);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}
