// ubench_mfma.hip -- what the f32 matrix pipe sustains by waves per SIMD (gfx950): bare v_mfma_f32_16x16x4_f32 streams, 16
// independent accumulators per wave, 4-wave workgroups (one wave per SIMD), 1 / 2 / 3 workgroups resident per CU (capped with
// unused dynamic LDS).  Built twice by tools/ubench_mfma.sh: accumulators in AGPRs (default) and in VGPRs
// (-mllvm -amdgpu-mfma-vgpr-form).  Prints TFLOP/s per configuration; the peak is 157.3 (MI355X_MICROARCH.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float seed) {
    extern __shared__ float dyn[];
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = seed + threadIdx.x * 0.001f + q; b[q] = seed - threadIdx.x * 0.002f + q; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(q + i) & 3], b[(q + 2 * i) & 3], acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (s[0] == 12345.678f) { out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3]; dyn[threadIdx.x] = s[0]; }
}

int main() {
    float* out;
    hipMalloc(&out, 4 << 20);
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    hipFuncSetAttribute((const void*)k_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 4000;
    for (int occ = 1; occ <= 4; ++occ) {
        const size_t lds = occ == 1 ? 82 * 1024 : occ == 2 ? 54 * 1024 : occ == 3 ? 41 * 1024 : 33 * 1024;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_mfma, dim3(ncu * occ), dim3(256), lds, 0, out, iters, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 5;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_mfma, dim3(ncu * occ), dim3(256), lds, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)ncu * occ * 4 * iters * 64.0 * 2048.0 * reps;
        printf("%d workgroup(s) of 4 waves per CU (%d wave(s) per SIMD): %.1f TFLOP/s (%.1f %% of 157.3)\n", occ, occ, flop / (ms * 1e-3) / 1e12,
               100.0 * flop / (ms * 1e-3) / 1e12 / 157.3);
    }
    return 0;
}
