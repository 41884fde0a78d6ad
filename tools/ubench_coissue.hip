// ubench_coissue.hip -- can exact-f32 VALU FMAs (v_fmac_f32 with a DPP row-rotate operand) run in the shadow of f32 MFMAs?
// One 4-wave workgroup per CU; per loop trip 32 v_mfma_f32_16x16x4_f32 (8 accumulators) and NV v_fmac_f32_dpp (32 accumulators),
// interleaved by sched_group_barrier.  Prints cycles-equivalent time per trip for NV = 0, 32, 64, 96, 128, 192.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// operand of lane (l + r) mod 16 of the same 16-lane row (DPP row_ror:r; the control must be a literal)
__device__ __forceinline__ float row_ror(float x, int r) {
    const int v = __builtin_bit_cast(int, x);
#define RR(n) case n: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x120 + n, 0xf, 0xf, true));
    switch (r) { RR(1) RR(2) RR(3) RR(4) RR(5) RR(6) RR(7) RR(8) RR(9) RR(10) RR(11) RR(12) RR(13) RR(14) RR(15) default: return x; }
#undef RR
}

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float va[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) va[i] = 0.f;
    float a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = seed + threadIdx.x * 0.001f + q; b[q] = seed - threadIdx.x * 0.002f + q; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(q + i) & 3], b[(q + 2 * i) & 3], acc[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int rot = (v & 15);
            // x operand of lane (l + rot) mod 16 within its row of 16 lanes (v_fmac_f32_dpp row_ror)
            const float xr = row_ror(b[(v >> 4) & 3], rot);
            va[v & 31] = __builtin_fmaf(a[(v >> 2) & 3], xr, va[v & 31]);
        }
        if (NV > 0) {
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV / 32, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) sv += va[i];
    if (s[0] + sv == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + sv;
}

template <int NV>
static void run(float* out, int ncu) {
    const int iters = 4000, reps = 5;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(ncu), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<NV>, dim3(ncu), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_trip = ms * 1e6 / (reps * (double)iters);
    printf("NV = %3d VALU FMAs beside 32 MFMAs: %.1f ns per trip (32 MFMAs alone = 1024 cycles = 427 ns at 2.4 GHz); MFMA rate %.1f TFLOP/s, VALU rate %.1f TFLOP/s\n",
           NV, ns_per_trip, ncu * 4 * 32 * 2048.0 / ns_per_trip / 1e3, ncu * 4 * NV * 128.0 / ns_per_trip / 1e3);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4 << 20);
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    run<0>(out, ncu); run<32>(out, ncu); run<64>(out, ncu); run<96>(out, ncu); run<128>(out, ncu); run<192>(out, ncu);
    return 0;
}
