// ubench_mfma_shape.hip -- does the SHAPE of the bf16 MFMA matter under the board's power cap?  (gfx950)
//
// The bf16 x 9 GEMMs sit at the 1400 W cap with the clock pulled down to ~1.87 GHz (profiles/r04_power_probe.txt).  Both bf16 MFMA
// shapes have the same rate (1024 flop / cycle / SIMD), but v_mfma_f32_32x32x16_bf16 reads HALF the operand registers per flop
// (32x16 + 32x16 elements for 32768 flop, against 16x32 + 16x32 for 16384 with v_mfma_f32_16x16x32_bf16).  This runs the bare
// nine-term loop of a 64 x 32 wave tile (registers only, random operand bits, no LDS, no memory) in both shapes, back to back for
// ~1.5 s each after a warm-up so that the clock settles, and prints useful fp32 TFLOP/s (= bf16 flop / 9) and the clock the time
// implies for a matrix pipe that never idles.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_shape.hip -o /tmp/ubench_mfma_shape && /tmp/ubench_mfma_shape
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 64 (w rows) x 32 (x rows) per wave as 4 x 2 fragments of 16 x 16, K = 32 per iteration: 9 terms x 8 = 72 MFMAs of 16 cycles
// SLEEP > 0: s_sleep SLEEP (64 cycles each) once per iteration of 1152 MFMA cycles -- a matrix pipe that idles on purpose: if the
// chip gives the same TFLOP/s back through a higher clock, idle cycles are free and the limit is power / current, not issue
template <int SLEEP>
__global__ __launch_bounds__(256) void k_16x16x32(const bf16x8* __restrict__ src, float* out, int iters) {
    extern __shared__ float dyn[];
    const int lane = threadIdx.x;
    bf16x8 fw[3][4], fx[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[p][i] = src[(p * 4 + i) * 256 + lane];
#pragma unroll
        for (int j = 0; j < 2; ++j) fx[p][j] = src[(12 + p * 2 + j) * 256 + lane];
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 2; p >= 0; --p)
#pragma unroll
            for (int q = 2; q >= 0; --q)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[p][i], fx[q][j], acc[i][j], 0, 0, 0);
        if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    f32x4 s = acc[0][0];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc[i][j];
    if (s[0] == 12345.678f) { out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3]; dyn[threadIdx.x] = s[0]; }
}

// the same wave tile as 2 x 1 fragments of 32 x 32, K = 32 per iteration as two halves of 16: 9 terms x 4 = 36 MFMAs of 32 cycles
// (NJ = 2: a 64 x 64 wave tile, four independent accumulators -- twice the flop per iteration -- to tell a dependency stall of the
//  two-accumulator form from a clock effect)
template <int NJ>
__global__ __launch_bounds__(256) void k_32x32x16(const bf16x8* __restrict__ src, float* out, int iters) {
    extern __shared__ float dyn[];
    const int lane = threadIdx.x;
    bf16x8 fw[3][2][2], fx[3][NJ][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) fw[p][i][h] = src[(p * 4 + i * 2 + h) * 256 + lane];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) fx[p][j][h] = src[((NJ == 1 ? 12 : 0) + j * 6 + p * 2 + h) * 256 + lane];
    }
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 2; p >= 0; --p)
#pragma unroll
            for (int q = 2; q >= 0; --q)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[p][i][h], fx[q][j][h], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) { out[blockIdx.x * 256 + threadIdx.x] = s; dyn[threadIdx.x] = s; }
}

template <typename KFN>
static void run(const char* name, KFN kern, const bf16x8* src, float* out, int ncu, int occ, int iters, double scale = 1.0) {
    const size_t lds = occ == 1 ? 82 * 1024 : 54 * 1024;        // caps the workgroups resident per CU
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.8) {      // the clock needs ~0.6 s to settle
        hipLaunchKernelGGL(kern, dim3(ncu * occ), dim3(256), lds, 0, src, out, iters);
        hipDeviceSynchronize();
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 40;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(ncu * occ), dim3(256), lds, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bf16_flop = scale * (double)ncu * occ * 4 * iters * 72.0 * 16384.0 * reps;
    const double cycles = scale * (double)occ * iters * 72.0 * 16.0 * reps;                // per SIMD, matrix pipe never idle
    printf("%-28s %d wave(s) per SIMD: %7.1f TFLOP/s of useful fp32 flop (bf16 flop / 9), %6.1f bf16 TFLOP/s, implied clock %.2f GHz, %.1f ms\n", name, occ,
           bf16_flop / 9.0 / (ms * 1e-3) / 1e12, bf16_flop / (ms * 1e-3) / 1e12, cycles / (ms * 1e-3) / 1e9, ms);
}

int main() {
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<unsigned short> h(18 * 256 * 8);
    srand(7);
    for (auto& v : h) {                                          // random bf16 in +-[0.5, 2): every mantissa and sign bit toggles
        const unsigned m = rand() & 0x7f, e = 126 + rand() % 2, sg = rand() & 1;
        v = (unsigned short)((sg << 15) | (e << 7) | m);
    }
    bf16x8* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 4 << 20);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 6000;
    for (int round = 0; round < 2; ++round)
        for (int occ = 1; occ <= 2; ++occ) {
            run("v_mfma_f32_16x16x32_bf16", k_16x16x32<0>, src, out, ncu, occ, iters);
            if (occ == 1) {
                run("  same + s_sleep 2 / iter", k_16x16x32<2>, src, out, ncu, occ, iters);
                run("  same + s_sleep 4 / iter", k_16x16x32<4>, src, out, ncu, occ, iters);
                run("  same + s_sleep 8 / iter", k_16x16x32<8>, src, out, ncu, occ, iters);
            }
            run("v_mfma_f32_32x32x16_bf16", k_32x32x16<1>, src, out, ncu, occ, iters);
            run("  same, 64x64 wave tile", k_32x32x16<2>, src, out, ncu, occ, iters / 2, 2.0);
        }
    return 0;
}
