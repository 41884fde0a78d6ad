// ubench_ldsdma.hip -- what the LDS of a CU takes per clock (gfx950): L2-resident 1 KiB pieces global -> LDS by LDS-DMA
// (global_load_lds_dwordx4), conflict-free ds_read_b128 sweeps, and both together, from 4 or 8 waves per CU.  Sizes the operand
// staging of the split-bf16 GEMM (tools/ubench_bf16x9.hip): bytes staged and bytes read per MFMA cycle.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsdma.hip -o tools/bin/ubench_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)

// MODE 0: DMA only; 1: reads only; 2: both (DMA pieces and reads interleaved 1 : RD)
template <int MODE, int RD, int FORM>
__global__ void k_lds(const f32x4* __restrict__ src, float* __restrict__ out, unsigned long long* clk, int iters, int src_pieces) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];       // 64 KiB ring: 64 pieces of 1 KiB
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 macc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) macc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ma, mb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ma[e] = (__bf16)(0.37f * (float)(lane + e + 1)); mb[e] = (__bf16)(1.0f / (float)(lane + 2 * e + 1)); }
    // every CU reads its own 64 KiB of the source (L2-resident after the first pass)
    const f32x4* ubase = src + (size_t)(blockIdx.x % src_pieces) * 64 * 64;      // wave-uniform (kernel argument + blockIdx)
    const f32x4* vbase = ubase + lane;                                            // per-lane 64-bit pointer
    const unsigned lane16 = lane * 16;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int piece = (wave * 8 + u) % 64;
            if (MODE != 1 && MODE != 4) {
                if (FORM == 0) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + piece * 64),
                                                     (__attribute__((address_space(3))) void*)&lds[piece * 64], 16, 0, 0);
                } else {
                    // SGPR base + 32-bit VGPR byte offset: global_load_lds_dwordx4 v_off, s[base:base+1]
                    const f32x4* pb = ubase + piece * 64;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(pb),
                                 "s"((unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)&lds[piece * 64]) : "memory");
                }
            }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int r = 0; r < RD; ++r) acc += lds[((piece + 8 * nw + r * 3) % 64) * 64 + lane];
            }
            if (MODE >= 3) {       // RD independent bf16 MFMAs (16 cycles each) behind the piece
#pragma unroll
                for (int r = 0; r < RD; ++r) macc[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, macc[r & 7], 0, 0, 0);
            }
        }
        if (MODE != 1 && MODE != 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 8; ++r) acc += macc[r];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int MODE, int RD, int FORM = 0>
static void run(const char* what, const f32x4* src, float* out, unsigned long long* clk, int ncu, int waves) {
    const int iters = 2000;
    auto kern = k_lds<MODE, RD, FORM>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(ncu), dim3(64 * waves), 96 * 1024, 0, src, out, clk, iters, 64);
    CK(hipDeviceSynchronize());
    unsigned long long h[1024];
    CK(hipMemcpy(h, clk, 8 * ncu, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (int i = 0; i < ncu; ++i) cyc += (double)h[i];
    cyc /= ncu;
    const double dma = (MODE != 1 && MODE != 4) ? (double)waves * 8 * 1024 * iters : 0, rd = (MODE == 1 || MODE == 2) ? (double)waves * 8 * RD * 1024 * iters : 0;
    printf("  %-34s %d waves/CU: %8.0f cycles = %6.1f per piece-slot | DMA %6.1f B/clk/CU  reads %6.1f B/clk/CU", what, waves, cyc, cyc / (8.0 * iters), dma / cyc, rd / cyc);
    if (MODE >= 3) printf("  | %d MFMAs per slot = %d cycles of matrix pipe", RD, 16 * RD);
    printf("\n");
}

int main() {
    int ncu = 256;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    f32x4* src; float* out; unsigned long long* clk;
    CK(hipMalloc(&src, 64 * 65536)); CK(hipMemset(src, 0, 64 * 65536)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&clk, 8 * 1024));
    printf("== LDS-DMA (L2-resident source, 1 KiB pieces) and ds_read_b128 rates per CU, every CU busy\n");
    for (int waves : {4, 8}) {
        run<0, 0>("LDS-DMA only", src, out, clk, ncu, waves);
        run<0, 0, 1>("LDS-DMA only, SGPR base + voffset", src, out, clk, ncu, waves);
        run<2, 2, 1>("DMA(saddr) + 2 reads per piece", src, out, clk, ncu, waves);
        run<1, 2>("ds_read_b128 only", src, out, clk, ncu, waves);
        run<2, 1>("DMA + 1 read per piece", src, out, clk, ncu, waves);
        run<2, 2>("DMA + 2 reads per piece", src, out, clk, ncu, waves);
        run<2, 4>("DMA + 4 reads per piece", src, out, clk, ncu, waves);
        run<4, 4>("4 MFMAs per slot, no DMA", src, out, clk, ncu, waves);
        run<3, 4>("DMA + 4 MFMAs per piece", src, out, clk, ncu, waves);
        run<4, 8>("8 MFMAs per slot, no DMA", src, out, clk, ncu, waves);
        run<3, 8>("DMA + 8 MFMAs per piece", src, out, clk, ncu, waves);
        run<3, 12>("DMA + 12 MFMAs per piece", src, out, clk, ncu, waves);
        run<3, 16>("DMA + 16 MFMAs per piece", src, out, clk, ncu, waves);
    }
    return 0;
}
