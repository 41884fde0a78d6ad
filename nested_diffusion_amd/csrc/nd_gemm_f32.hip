// nd_gemm_f32.hip -- fp32 Linear layers of the ViT blocks (timm 0.4.12 Attention.qkv / proj, Mlp.fc1 / fc2, PatchEmbed.proj as a
// GEMM over im2col rows; call sites classification_train_separately.py:337-346):
//     out[m,n] = act(sum_k x[m,k] w[n,k] + bias[n]) + res[m,n],   x [M][K] and w [N][K] row-major, K-contiguous.
// gfx950 only; built with -mllvm -amdgpu-mfma-vgpr-form (accumulators in VGPRs: nested_diffusion_amd/build.py).
//
// Workgroup tile BM x BN (256 threads = 2 x 2 waves), K-step 16, v_mfma_f32_16x16x4_f32 (exact f32).  Three-stage pipeline,
// ONE barrier per K-step: while the 32 MFMAs of step k run on register set k & 1,
//     * the global tile of step k+1 (requested during step k-1, in staging registers) is written to LDS buffer (k+1) & 1,
//     * the global loads of step k+2 are issued into the freed staging registers,
//     * (barrier) the fragments of step k+1 are read from LDS into the other register set,
// dealt between the MFMAs by sched_group_barrier, so neither the global latency nor the LDS latency is ever waited for and
// every memory instruction issues in the shadow of running MFMAs.
// LDS rows are 16 floats + 8 pad: a lane's float4 (k = 4*(l>>4)..+3) is one conflict-free ds_read_b128; MFMA jj takes element jj
// of every lane (k order permuted identically on both operands).
#include "nd_common.hpp"

#define GB_K 16
#define GB_LD 24

template <int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm_nt(const float* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ bias, const float* __restrict__ res,
                                                 float* __restrict__ out, int M, int K, int N, int act, int n_full,
                                                 int split, float* __restrict__ part) {
    constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
    constexpr int FM = WM / 16, FN = WN / 16;    // 16x16 fragments per wave
    constexpr int LA = BM * GB_K / 4 / 256;      // float4 loads per thread for the x tile
    constexpr int LB = BN * GB_K / 4 / 256;
    static_assert(LA >= 1 && LB >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) float sA[2][BM][GB_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][BN][GB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order: consecutive tiles along N (sharing the x panel) land on one XCD.
    // Workgroups [0, n_full) take one whole tile each; the remaining tiles are cut into `split` k-slabs, one workgroup per
    // slab, whose raw sums go to `part` and are finished by k_gemm_fixup (the tail of the launch is then `split` times finer).
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

    // staging map: thread -> (row, kq), 4 threads per 16-float row; the row pointers are fixed for the whole tile
    const float* pa[LA];
    const float* pb[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) { const int e = tid + i * 256; pa[i] = x + (size_t)min(m0 + (e >> 2), M - 1) * K + (e & 3) * 4; }
#pragma unroll
    for (int i = 0; i < LB; ++i) { const int e = tid + i * 256; pb[i] = w + (size_t)min(n0 + (e >> 2), N - 1) * K + (e & 3) * 4; }
    f32x4 ra[LA], rb[LB];                              // staging registers: one K-step of both operands
    f32x4 fa[2][FM], fb[2][FN];                        // two fragment sets, indexed by step parity (static after unrolling)
#define GB_GLOAD(k0)                                                                                                       \
    {                                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(pa[i] + (k0));             \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(pb[i] + (k0));             \
    }
#define GB_SWRITE(buf)                                                                                                     \
    {                                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) { const int e = tid + i * 256; *reinterpret_cast<f32x4*>(&sA[buf][e >> 2][(e & 3) * 4]) = ra[i]; } \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) { const int e = tid + i * 256; *reinterpret_cast<f32x4*>(&sB[buf][e >> 2][(e & 3) * 4]) = rb[i]; } \
    }
#define GB_READ(set, buf)                                                                                                  \
    {                                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(&sA[buf][wr * WM + 16 * i + lr][lk]); \
        _Pragma("unroll") for (int j = 0; j < FN; ++j) fb[set][j] = *reinterpret_cast<const f32x4*>(&sB[buf][wc * WN + 16 * j + lr][lk]); \
    }
    // k-quads q0 .. q1-1 of one step: FM*FN independent accumulators between two MFMAs on the same one
#define GB_MMA(set, q0, q1)                                                                                                \
    {                                                                                                                      \
        _Pragma("unroll") for (int q = (q0); q < (q1); ++q)                                                               \
            _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                                \
                _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                            \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[set][j][q], fa[set][i][q], acc[i][j], 0, 0, 0);   \
    }
    const int nkt = K / GB_K;
    const int ks0 = slab < 0 ? 0 : (int)((long)slab * nkt / split);
    const int nk = slab < 0 ? nkt : (int)((long)(slab + 1) * nkt / split);
    const int lr = lane & 15, lk = 4 * (lane >> 4);
    constexpr int NMQ = FM * FN;                       // MFMAs per k-quad
    constexpr int NW = LA + LB, NR = FM + FN;          // LDS writes / global loads, and fragment reads, per step

    // prologue: step ks0 -> LDS buffer 0 -> set 0; step ks0+1 requested
    GB_GLOAD(ks0 * GB_K)
    GB_SWRITE(0)
    __syncthreads();
    GB_GLOAD(min(ks0 + 1, nk - 1) * GB_K)
    GB_READ(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    for (int ks = ks0; ks < nk; ks += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ks + u < nk) {
                // first half: 2 k-quads of step ks+u || step ks+u+1 (staging registers) -> LDS buffer (u+1)&1 || request step ks+u+2
                GB_SWRITE((u + 1) & 1)
                GB_GLOAD(min(ks + u + 2, nk - 1) * GB_K)
                GB_MMA(u, 0, 2)
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NMQ - 4 * NW > 0 ? 2 * NMQ - 4 * NW : 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                       // step ks+u+1 is in LDS for everybody; its buffer's old readers are done
                // second half: the other 2 k-quads || fragments of step ks+u+1 -> the other register set
                GB_READ((u + 1) & 1, (u + 1) & 1)
                GB_MMA(u, 2, 4)
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NMQ - 2 * NR > 0 ? 2 * NMQ - 2 * NR : 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#undef GB_GLOAD
#undef GB_SWRITE
#undef GB_READ
#undef GB_MMA
    // D[n = 4*(l>>4)+r][m = l&15]: a lane owns 4 consecutive n of one row m
    if (slab >= 0) {
        float* pt = part + ((size_t)(bid - n_full) * split + slab) * (BM * BN);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<f32x4*>(pt + (wr * WM + 16 * i + (lane & 15)) * BN + wc * WN + 16 * j + 4 * (lane >> 4)) = acc[i][j];
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

// 128 x 64 tiles (GT_BM x GT_BN of nd_vit.hip); grid = whole tiles + remainder tiles x split
hipError_t nd_launch_gemm_nt_128x64(const float* x, const float* w, const float* bias, const float* res, float* out, int M, int K, int N,
                                    int act, int n_full, int split, float* part, unsigned grid, hipStream_t st) {
    hipLaunchKernelGGL((k_gemm_nt<128, 64>), dim3(grid), dim3(256), 0, st, x, w, bias, res, out, M, K, N, act, n_full, split, part);
    return hipGetLastError();
}
