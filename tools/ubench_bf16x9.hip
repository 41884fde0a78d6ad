// ubench_bf16x9.hip -- gate for "exact f32 products off the 1/16-rate pipe" (gfx950).
//
// An fp32 value a splits EXACTLY into three bf16 pieces a1 = rn(a), a2 = rn(a - a1), a3 = a - a1 - a2 (8 + 8 + 8 significand
// bits); the nine piece products a_p b_q are exact in fp32, and v_mfma_f32_16x16x32_bf16 accumulates 32 of them per 16 cycles where
// v_mfma_f32_16x16x4_f32 takes 32 cycles for 4 exact-f32 products: 9/16 of the matrix-pipe CYCLES per product.  Whether that is
// 9/16 of the TIME is what this file measures, on random data (the chip clocks bf16 MFMA loops lower than f32 ones):
//   A. bare loops, operands in registers, one wave per SIMD: f32 chain vs 9-term bf16 chain, TFLOP/s-equivalent + in-kernel clock
//   B. a whole GEMM out[m,n] = sum_k x[m,k] w[n,k] on PRE-SPLIT operands ("frag32b3": per (16 rows x 32 k) block three 1 KiB
//      planes in MFMA lane order; global -> LDS by LDS-DMA, LDS -> registers by ds_read_b128), LDS ring of NS slots, two register
//      sets, one barrier per K-step; shapes: the ViT qkv GEMM (M 6272, K 768, N 2304) and the mc_trials = 20 ConditionalLinear
//      block (5 members x [640 x 4096] x [4096 x 4096]); TFLOP/s-equivalent WITH and without the activation split pass, and the
//      error against an fp64 product of the same fp32 inputs (library kernels today: 9.7e-7 by the same measure).
// Yardsticks (library, rocprofv3): k_gemm_nt qkv 192 us = 116 TFLOP/s; k_cond_gemm 800-810 us per launch = 133 TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_bf16x9.hip -o /tmp/ubench_bf16x9
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(e)                                                                                   \
    do {                                                                                        \
        hipError_t _e = (e);                                                                    \
        if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } \
    } while (0)

// ------------------------------------------------------------------------------------------------ A. bare loops
__device__ __forceinline__ unsigned xs(unsigned& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ __launch_bounds__(256) void k_bare_f32(float* out, unsigned long long* clk, int iters, unsigned seed) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    float a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = (float)(xs(s) & 0xffffff) * (1.f / 8388608.f) - 1.f; b[q] = (float)(xs(s) & 0xffffff) * (1.f / 8388608.f) - 1.f; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(q + i) & 3], b[(q + 2 * i) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 t = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) t += acc[i];
    if (t[0] == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k_bare_b9(float* out, unsigned long long* clk, int iters, unsigned seed) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    bf16x8 fa[3][4], fb[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                fa[p][i][e] = (__bf16)(((float)(xs(s) & 0xffff) * (1.f / 32768.f) - 1.f) * (p == 0 ? 1.f : p == 1 ? 0.004f : 0.000015f));
                fb[p][i][e] = (__bf16)(((float)(xs(s) & 0xffff) * (1.f / 32768.f) - 1.f) * (p == 0 ? 1.f : p == 1 ? 0.004f : 0.000015f));
            }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 2; p >= 0; --p)
#pragma unroll
            for (int q = 2; q >= 0; --q)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[p][i], fb[q][j], acc[i * 4 + j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 t = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) t += acc[i];
    if (t[0] == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// ------------------------------------------------------------------------------------------------ B. split + GEMM
// x [R][K] row-major fp32 -> frag32b3: block (r/16, k/32) = 3 planes x 64 lanes x 8 bf16; lane l = row l&15, k = 8*(l>>4)..+7.
// One wave per block: a lane reads 32 contiguous bytes of its row, writes 16 B into each plane (three coalesced 1 KiB stores).
__global__ __launch_bounds__(256) void k_split3(const float* __restrict__ x, bf16x8* __restrict__ out, int R, int K, int Rpad) {
    const int lane = threadIdx.x & 63;
    const long blk = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nkb = K / 32;
    if (blk >= (long)(Rpad / 16) * nkb) return;
    const int rb = (int)(blk / nkb), kb = (int)(blk % nkb);
    const int row = rb * 16 + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
    float v[8];
    if (row < R) {
        const float4 u0 = *reinterpret_cast<const float4*>(x + (size_t)row * K + k0), u1 = *reinterpret_cast<const float4*>(x + (size_t)row * K + k0 + 4);
        v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    bf16x8 h1, h2, h3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h1[e] = (__bf16)v[e];
        const float r1 = v[e] - (float)h1[e];
        h2[e] = (__bf16)r1;
        h3[e] = (__bf16)(r1 - (float)h2[e]);
    }
    out[(blk * 3 + 0) * 64 + lane] = h1;
    out[(blk * 3 + 1) * 64 + lane] = h2;
    out[(blk * 3 + 2) * 64 + lane] = h3;
}

// Workgroup tile (2*FA*16 rows of w) x (2*FB*16 rows of x), 2 x 2 waves of FA x FB fragments; K-step 32.
// NT = 9: all piece products; NT = 6: without a2 b3, a3 b2, a3 b3 (<= 2^-24 of the product each) -- labelled, not the "exact" form.
template <int FA, int FB, int NS, int NT>
__global__ __launch_bounds__(256) void k_gemm_b9(const bf16x8* __restrict__ wP, const bf16x8* __restrict__ xP, float* __restrict__ out,
                                                 int M, int N, int K, long wStride, long xStride, int TM, int TN, int nwg,
                                                 unsigned long long* __restrict__ clk) {
    constexpr int NFRAG = 2 * FA + 2 * FB;          // fragments per slot
    constexpr int NP = NFRAG * 3 / 4;               // LDS-DMA pieces per wave per step
    static_assert((NFRAG * 3) % 4 == 0, "pieces must deal evenly to 4 waves");
    extern __shared__ __attribute__((aligned(16))) bf16x8 lds[];   // [NS][NFRAG][3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    int bid = blockIdx.x;
    {   // blocks b and b + 8 share an XCD: XCD x takes a contiguous run of tiles
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int per = TM * TN, member = bid / per, r2 = bid - member * per, tn = r2 / TM, tm = r2 - tn * TM;
    const int nkb = K / 32, nfr = (N + 15) >> 4, mfr = (M + 15) >> 4;
    const bf16x8* wb = wP + (size_t)member * wStride;
    const bf16x8* xb = xP + (size_t)member * xStride;
    const bf16x8* src[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int e = wave * NP + u, f = e / 3, pl = e % 3;
        const bf16x8* base = f < 2 * FA ? wb + (size_t)min(tn * 2 * FA + f, nfr - 1) * nkb * 192 : xb + (size_t)min(tm * 2 * FB + f - 2 * FA, mfr - 1) * nkb * 192;
        src[u] = base + pl * 64 + lane;
    }
#define B9_STAGE(slot, step)                                                                                                        \
    {                                                                                                                               \
        _Pragma("unroll") for (int pc_ = 0; pc_ < NP; ++pc_)                                                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[pc_] + (size_t)(step) * 192),      \
                                             (__attribute__((address_space(3))) void*)&lds[((slot) * NFRAG * 3 + wave * NP + pc_) * 64], 16, 0, 0); \
    }
#define B9_READ(set, slot)                                                                                                          \
    {                                                                                                                               \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                             \
            _Pragma("unroll") for (int i = 0; i < FA; ++i) fw[set][p][i] = lds[(((slot) * NFRAG + wn * FA + i) * 3 + p) * 64 + lane]; \
            _Pragma("unroll") for (int j = 0; j < FB; ++j) fx[set][p][j] = lds[(((slot) * NFRAG + 2 * FA + wm * FB + j) * 3 + p) * 64 + lane]; \
        }                                                                                                                           \
    }
#define B9_TERM(set, p, q)                                                                                                          \
    {                                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < FA; ++i)                                                                              \
            _Pragma("unroll") for (int j = 0; j < FB; ++j)                                                                          \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[set][p][i], fx[set][q][j], acc[i][j], 0, 0, 0);              \
    }
#define B9_SYNC()                                                                                                                   \
    {                                                                                                                               \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP * (NS - 2)) : "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
    }
    f32x4 acc[FA][FB];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fw[2][3][FA], fx[2][3][FB];
    const int nk = nkb;
#pragma unroll
    for (int u = 0; u < NS; ++u) B9_STAGE(u, min(u, nk - 1))
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    B9_READ(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tc0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    constexpr int U = (NS % 2 == 0) ? NS : 2 * NS;
    constexpr int NM = NT * FA * FB, NMEM = NP + 3 * (FA + FB), RATIO = NM / NMEM > 0 ? NM / NMEM : 1;
    for (int s = 0; s < nk; s += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s + u < nk) {
                B9_SYNC()
                B9_STAGE(u % NS, min(s + u + NS, nk - 1))
                B9_READ((u + 1) & 1, (u + 1) % NS)
                if (NT == 9) { B9_TERM(u & 1, 2, 2) B9_TERM(u & 1, 2, 1) B9_TERM(u & 1, 1, 2) }
                B9_TERM(u & 1, 2, 0) B9_TERM(u & 1, 0, 2) B9_TERM(u & 1, 1, 1) B9_TERM(u & 1, 1, 0) B9_TERM(u & 1, 0, 1) B9_TERM(u & 1, 0, 0)
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int k = 0; k < 3 * (FA + FB); ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NM - RATIO * NMEM, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (clk && tid == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - tc0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - tr0; }
#undef B9_STAGE
#undef B9_READ
#undef B9_TERM
#undef B9_SYNC
    float* o = out + (size_t)member * M * N;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int n = (tn * 2 * FA + wn * FA + i) * 16 + 4 * (lane >> 4), m = (tm * 2 * FB + wm * FB + j) * 16 + (lane & 15);
            if (m < M && n + 3 < N) *reinterpret_cast<f32x4*>(o + (size_t)m * N + n) = acc[i][j];
        }
}


// HBM streaming filler for the "mixed load" measurement: what the chip clocks a GEMM at when it runs between memory-bound kernels
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ p, float* __restrict__ out, size_t n) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += __builtin_nontemporal_load(p + i);
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[threadIdx.x] = a[0];
}

// ---- second form: any wave grid (WN x WM waves of FA x FB fragments), k-split tail, two issue schedules -------------------------
// SCHED 0: memory instructions dealt evenly between the MFMAs of a step.  SCHED 1: the fragment reads of step j+1 first (one per 2
// MFMAs), then the LDS-DMA pieces of step j+NS (one per 4), and the last MFMAs of the step with nothing between them, so that the
// counted wait + barrier at the top of the next step find every read retired.
template <int FA, int FB, int WN, int WM, int NS, int SCHED>
__global__ __launch_bounds__(64 * WN * WM) void k_gemm_b9v2(const bf16x8* __restrict__ wP, const bf16x8* __restrict__ xP, float* __restrict__ out,
                                                            int M, int N, int K, long wStride, long xStride, int TM, int TN, int n_full, int split,
                                                            f32x4* __restrict__ part, unsigned long long* __restrict__ clk, int stagger_ticks) {
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
    // EXPERIMENT: de-phase the two workgroups that share a CU (they start together and then run in lockstep, so their prologues and
    // epilogues -- during which a workgroup issues no MFMA -- coincide): the second residency wave of the initial dispatch waits
    // EXPERIMENT (round 5, UB_PRIO): wave priorities (s_setprio) against the oldest-first issue arbitration that lets one of the two
    // co-resident workgroups run at its solo rate while the other only fills gaps.  stagger_ticks < 0 selects a scheme:
    //   -1  by SIMD pair: a workgroup's waves on SIMDs {0,1} are high when it arrived on its CU as an even arrival, on {2,3} when odd
    //   -2  finish boost: a wave raises its priority for the last quarter of its tile's K-steps
    //   -3  alternating: priority flips every K-step, in opposite phase for even and odd arrivals
    //   -4  start boost: high priority for the first quarter of the K-steps (a fresh tile catches up with its partner)
    unsigned arrival = 0;
    if (stagger_ticks < 0) {
        __shared__ unsigned s_arr;
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));          // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;    // HW_REG_XCC_ID
            const unsigned key = xcc * 256 + ((hw >> 13) & 7) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15);
            s_arr = atomicAdd(reinterpret_cast<unsigned*>(clk) + 2 * 65536 + key, 1u);
        }
        __syncthreads();
        arrival = __builtin_amdgcn_readfirstlane(s_arr);
        if (stagger_ticks == -1) {
            const unsigned simd = (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) >> 4) & 3u;
            if ((((simd >> 1) ^ arrival) & 1u) == 0) __builtin_amdgcn_s_setprio(2);
        }
    }
    if (stagger_ticks > 0 && blockIdx.x < 512) {
        // every other ARRIVAL on a CU waits (per-CU arrival counters keyed by XCC_ID / SE_ID / CU_ID, never reset: parity alternates)
        __shared__ unsigned s_order;
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));          // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;    // HW_REG_XCC_ID
            const unsigned key = xcc * 256 + ((hw >> 13) & 7) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15);
            s_order = atomicAdd(reinterpret_cast<unsigned*>(clk) + 2 * 65536 + key, 1u);
        }
        __syncthreads();
        if (s_order & 1) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)stagger_ticks) __builtin_amdgcn_s_sleep(8);
        }
    }
    constexpr int NW = WN * WM, NFRAG = WN * FA + WM * FB, NPC = NFRAG * 3;
    constexpr int NP = (NPC + NW - 1) / NW;                       // LDS-DMA pieces per wave per step (the last wave may have fewer)
    // (where NPC % NW != 0 the last waves re-issue the last piece: same bytes to the same place, every wave counts NP per step)
    extern __shared__ __attribute__((aligned(16))) bf16x8 lds[];   // [NS][NFRAG][3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WM, wm = wave % WM;
    int bid = blockIdx.x, slab = -1, rem_index = 0;
    if (bid < n_full) {
        const int q = n_full / 8, r = n_full % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    } else {
        const int j = bid - n_full;
        rem_index = j / split; slab = j % split; bid = n_full + rem_index;
    }
    const int per = TM * TN, member = bid / per, r2 = bid - member * per, tn = r2 / TM, tm = r2 - tn * TM;
    const int nkb = K / 32, nfr = (N + 15) >> 4, mfr = (M + 15) >> 4;
    const int c0 = slab < 0 ? 0 : (int)((long)slab * nkb / split), c1 = slab < 0 ? nkb : (int)((long)(slab + 1) * nkb / split);
    const int nk = c1 - c0;
    const bf16x8* wb = wP + (size_t)member * wStride;
    const bf16x8* xb = xP + (size_t)member * xStride;
    const bf16x8* src[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int e = min(wave * NP + u, NPC - 1), f = e / 3, pl = e % 3;
        const bf16x8* base = f < WN * FA ? wb + ((size_t)min(tn * WN * FA + f, nfr - 1) * nkb + c0) * 192
                                         : xb + ((size_t)min(tm * WM * FB + f - WN * FA, mfr - 1) * nkb + c0) * 192;
        src[u] = base + pl * 64 + lane;
    }
#define V2_STAGE(slot, step)                                                                                                        \
    {                                                                                                                               \
        _Pragma("unroll") for (int pc_ = 0; pc_ < NP; ++pc_)                                                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[pc_] + (size_t)(step) * 192),      \
                                             (__attribute__((address_space(3))) void*)&lds[((slot) * NPC + min(wave * NP + pc_, NPC - 1)) * 64], 16, 0, 0); \
    }
#define V2_READ(set, slot)                                                                                                          \
    {                                                                                                                               \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                             \
            _Pragma("unroll") for (int i = 0; i < FA; ++i) fw[set][p][i] = lds[(((slot) * NFRAG + wn * FA + i) * 3 + p) * 64 + lane]; \
            _Pragma("unroll") for (int j = 0; j < FB; ++j) fx[set][p][j] = lds[(((slot) * NFRAG + WN * FA + wm * FB + j) * 3 + p) * 64 + lane]; \
        }                                                                                                                           \
    }
#define V2_TERM(set, p, q)                                                                                                          \
    {                                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < FA; ++i)                                                                              \
            _Pragma("unroll") for (int j = 0; j < FB; ++j)                                                                          \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[set][p][i], fx[set][q][j], acc[i][j], 0, 0, 0);              \
    }
#define V2_SYNC()                                                                                                                   \
    {                                                                                                                               \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP * (NS - 2)) : "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
    }
    f32x4 acc[FA][FB];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fw[2][3][FA], fx[2][3][FB];
#pragma unroll
    for (int u = 0; u < NS; ++u) V2_STAGE(u, min(u, nk - 1))
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    V2_READ(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tc0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    constexpr int U = (NS % 2 == 0) ? NS : 2 * NS;
    constexpr int NM = 9 * FA * FB, NRD = 3 * (FA + FB), NMEM = NP + NRD, RATIO = NM / NMEM > 0 ? NM / NMEM : 1;
    for (int s = 0; s < nk; s += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s + u < nk) {
                if (stagger_ticks < -1) {
                    const int step = s + u;
                    if (stagger_ticks == -2) { if (step == nk - nk / 4) __builtin_amdgcn_s_setprio(3); }
                    else if (stagger_ticks == -3) { if ((step + (int)arrival) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
                    else if (stagger_ticks == -4) { if (step == 0) __builtin_amdgcn_s_setprio(3); else if (step == nk / 4) __builtin_amdgcn_s_setprio(0); }
                }
                V2_SYNC()
                if (SCHED == 0) { V2_STAGE(u % NS, min(s + u + NS, nk - 1)) V2_READ((u + 1) & 1, (u + 1) % NS) }
                else { V2_READ((u + 1) & 1, (u + 1) % NS) V2_STAGE(u % NS, min(s + u + NS, nk - 1)) }
                V2_TERM(u & 1, 2, 2) V2_TERM(u & 1, 2, 1) V2_TERM(u & 1, 1, 2)
                V2_TERM(u & 1, 2, 0) V2_TERM(u & 1, 0, 2) V2_TERM(u & 1, 1, 1) V2_TERM(u & 1, 1, 0) V2_TERM(u & 1, 0, 1) V2_TERM(u & 1, 0, 0)
                if (SCHED == 0) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
#pragma unroll
                    for (int k = 0; k < NRD; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NM - RATIO * NMEM, 0);
                } else {
                    constexpr int R1 = (2 * NRD + 3 * NP <= NM - 8) ? 2 : 1, R2 = (R1 * NRD + 3 * NP <= NM - 8) ? 3 : ((R1 * NRD + 2 * NP <= NM - 4) ? 2 : 1);
#pragma unroll
                    for (int k = 0; k < NRD; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, R1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, R2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NM - R1 * NRD - R2 * NP, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (clk && tid == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - tc0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - tr0; }
#undef V2_STAGE
#undef V2_READ
#undef V2_TERM
#undef V2_SYNC
    if (slab >= 0) {     // raw accumulators of a k-slab: [remainder tile][slab][wave][fragment][lane], 1 KiB per fragment
        f32x4* pt = part + (((size_t)rem_index * split + slab) * NW + wave) * (FA * FB) * 64 + lane;
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j) pt[(i * FB + j) * 64] = acc[i][j];
        return;
    }
    float* o = out + (size_t)member * M * N;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int n = ((tn * WN + wn) * FA + i) * 16 + 4 * (lane >> 4), m = ((tm * WM + wm) * FB + j) * 16 + (lane & 15);
            if (m < M && n + 3 < N) *reinterpret_cast<f32x4*>(o + (size_t)m * N + n) = acc[i][j];
        }
    if (clk && tid == 0) { clk[3 * 65536 + 2 * blockIdx.x] = t_entry; clk[3 * 65536 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

// sums the k-slabs of the remainder tiles in slab order and stores them: one wave per (remainder tile, wave sub-tile)
template <int FA, int FB, int WN, int WM>
__global__ __launch_bounds__(64) void k_b9_fixup(const f32x4* __restrict__ part, float* __restrict__ out, int M, int N, int TM, int TN, int n_full,
                                                 int split) {
    constexpr int NW = WN * WM;
    const int lane = threadIdx.x, wave = blockIdx.x % NW, ri = blockIdx.x / NW;
    const int wn = wave / WM, wm = wave % WM;
    const int bid = n_full + ri, per = TM * TN, member = bid / per, r2 = bid - member * per, tn = r2 / TM, tm = r2 - tn * TM;
    float* o = out + (size_t)member * M * N;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            f32x4 a = part[(((size_t)ri * split + 0) * NW + wave) * (FA * FB) * 64 + (i * FB + j) * 64 + lane];
            for (int k = 1; k < split; ++k) a += part[(((size_t)ri * split + k) * NW + wave) * (FA * FB) * 64 + (i * FB + j) * 64 + lane];
            const int n = ((tn * WN + wn) * FA + i) * 16 + 4 * (lane >> 4), m = ((tm * WM + wm) * FB + j) * 16 + (lane & 15);
            if (m < M && n + 3 < N) *reinterpret_cast<f32x4*>(o + (size_t)m * N + n) = a;
        }
}

static unsigned g_rng = 12345u;
static float frand() { g_rng = g_rng * 1664525u + 1013904223u; return (float)((g_rng >> 8) & 0xffff) / 32768.f - 1.f + (float)(g_rng >> 24) * 1e-6f; }

struct Shape { const char* name; int batch, M, K, N; double yard_us; };

template <int FA, int FB, int NS, int NT>
static void run_gemm(const Shape& sh, const float* dW, const float* dX, bf16x8* wP, bf16x8* xP, float* dOut, const std::vector<float>& hW,
                     const std::vector<float>& hX, int ncu) {
    const int BM = 2 * FB * 16, BN = 2 * FA * 16;
    const int TM = (sh.M + BM - 1) / BM, TN = (sh.N + BN - 1) / BN, nwg = sh.batch * TM * TN;
    const int Mpad = TM * BM, Npad = TN * BN;
    const size_t lds_bytes = (size_t)NS * (2 * FA + 2 * FB) * 3 * 1024;
    auto kern = k_gemm_b9<FA, FB, NS, NT>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds_bytes));
    const long wStride = (long)(Npad / 16) * (sh.K / 32) * 192, xStride = (long)(Mpad / 16) * (sh.K / 32) * 192;
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    // weights split once (not timed: done at load); activations split per GEMM (timed)
    for (int b = 0; b < sh.batch; ++b) {
        const long nb = (long)(Npad / 16) * (sh.K / 32);
        hipLaunchKernelGGL(k_split3, dim3((unsigned)((nb * 64 + 255) / 256)), dim3(256), 0, 0, dW + (size_t)b * sh.N * sh.K, wP + b * wStride, sh.N, sh.K, Npad);
    }
    auto split_x = [&]() {
        for (int b = 0; b < sh.batch; ++b) {
            const long nb = (long)(Mpad / 16) * (sh.K / 32);
            hipLaunchKernelGGL(k_split3, dim3((unsigned)((nb * 64 + 255) / 256)), dim3(256), 0, 0, dX + (size_t)b * sh.M * sh.K, xP + b * xStride, sh.M, sh.K, Mpad);
        }
    };
    static unsigned long long* dclk = nullptr;
    if (!dclk) CK(hipMalloc(&dclk, 16 * 16384));
    auto gemm = [&]() { hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds_bytes, 0, wP, xP, dOut, sh.M, sh.N, sh.K, wStride, xStride, TM, TN, nwg, dclk); };
    split_x(); gemm(); gemm();
    CK(hipDeviceSynchronize());
    // the chip's clock under load settles over hundreds of milliseconds: keep the kernel running for ~0.6 s before timing it
    {
        hipEvent_t w0, w1;
        CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
        CK(hipEventRecord(w0)); gemm(); CK(hipEventRecord(w1)); CK(hipEventSynchronize(w1));
        float one = 0; CK(hipEventElapsedTime(&one, w0, w1));
        const int nwarm = (int)(600.0f / (one > 0.01f ? one : 0.01f));
        for (int r = 0; r < nwarm; ++r) gemm();
        CK(hipDeviceSynchronize());
    }
    const int reps = 200;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) gemm();
    CK(hipEventRecord(e1));
    for (int r = 0; r < reps; ++r) split_x();
    CK(hipEventRecord(e2));
    CK(hipEventSynchronize(e2));
    float ms_g = 0, ms_s = 0;
    CK(hipEventElapsedTime(&ms_g, e0, e1)); CK(hipEventElapsedTime(&ms_s, e1, e2));
    const double us_g = ms_g * 1e3 / reps, us_s = ms_s * 1e3 / reps;
    const double flop = 2.0 * sh.batch * sh.M * (double)sh.N * sh.K;
    // error against fp64 on sampled rows of member 0 and of the last member
    std::vector<float> hOut((size_t)sh.batch * sh.M * sh.N);
    CK(hipMemcpy(hOut.data(), dOut, hOut.size() * 4, hipMemcpyDeviceToHost));
    double maxref = 0, maxerr = 0, se = 0, sr = 0;
    const int bs[2] = {0, sh.batch - 1};
    for (int bi = 0; bi < (sh.batch > 1 ? 2 : 1); ++bi) {
        const int b = bs[bi];
        for (int mi = 0; mi < 24; ++mi) {
            const int m = (int)(((long)mi * 2654435761u) % sh.M);
            for (int n = 0; n < sh.N; n += 3) {
                double a = 0;
                const float* xr = &hX[((size_t)b * sh.M + m) * sh.K];
                const float* wr = &hW[((size_t)b * sh.N + n) * sh.K];
                for (int k = 0; k < sh.K; ++k) a += (double)xr[k] * (double)wr[k];
                const double d = fabs((double)hOut[((size_t)b * sh.M + m) * sh.N + n] - a);
                maxref = fmax(maxref, fabs(a)); maxerr = fmax(maxerr, d); se += d * d; sr += a * a;
            }
        }
    }
    if (getenv("UB_DEBUG")) {
        std::vector<unsigned short> hp((size_t)xStride * 8);
        CK(hipMemcpy(hp.data(), xP, hp.size() * 2, hipMemcpyDeviceToHost));
        auto bf = [](unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
        long bad = 0;
        for (int r = 0; r < sh.M; ++r)
            for (int k = 0; k < sh.K; ++k) {
                const size_t blk = (size_t)(r / 16) * (sh.K / 32) + k / 32;
                const int lane = (r % 16) + 16 * ((k % 32) / 8), e = k % 8;
                const float s3 = bf(hp[((blk * 3 + 0) * 64 + lane) * 8 + e]) + bf(hp[((blk * 3 + 1) * 64 + lane) * 8 + e]) + bf(hp[((blk * 3 + 2) * 64 + lane) * 8 + e]);
                if (s3 != hX[(size_t)r * sh.K + k]) ++bad;
            }
        long nonfinite = 0;
        for (float v : hOut) if (!std::isfinite(v)) ++nonfinite;
        printf("    [debug] split planes that do not sum back exactly: %ld of %ld; non-finite outputs %ld of %zu; out[0..3] = %g %g %g %g\n", bad,
               (long)sh.M * sh.K, nonfinite, hOut.size(), hOut[0], hOut[1], hOut[2], hOut[3]);
    }
    std::vector<unsigned long long> hc(2 * (size_t)nwg);
    CK(hipMemcpy(hc.data(), dclk, 16 * (size_t)nwg, hipMemcpyDeviceToHost));
    double cyc = 0, ghz = 0;
    for (int b = 0; b < nwg; ++b) { cyc += (double)hc[2 * b]; ghz += (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1; }
    cyc /= nwg; ghz /= nwg;
    const double ideal = (double)NT * FA * FB * 16.0;
    printf("    main loop: %.0f cycles per K-step per workgroup (MFMA-only %.0f x %d resident = %.0f; ratio %.2f), in-kernel clock %.2f GHz\n",
           cyc / (sh.K / 32), ideal, occ, ideal * occ, cyc / (sh.K / 32) / (ideal * occ), ghz);
    const double rounds = (double)nwg / (ncu * (occ > 0 ? occ : 1));
    printf("  tile %3dx%-3d NS=%d terms=%d  %d wg/CU  %5d wgs (%.2f rounds)  gemm %7.1f us = %6.1f TF-eq | + x split %5.1f us -> %6.1f TF-eq"
           " | vs library %.0f us: %.2fx (%.2fx incl. split) | err max/maxref %.2e rms %.2e\n",
           BN, BM, NS, NT, occ, nwg, rounds, us_g, flop / us_g * 1e-6, us_s, flop / (us_g + us_s) * 1e-6, sh.yard_us, sh.yard_us / us_g,
           sh.yard_us / (us_g + us_s), maxerr / maxref, sqrt(se / sr));
    fflush(stdout);
}


template <int FA, int FB, int WN, int WM, int NS, int SCHED>
static void run_gemm2(const Shape& sh, const float* dW, const float* dX, bf16x8* wP, bf16x8* xP, float* dOut, const std::vector<float>& hW,
                      const std::vector<float>& hX, int ncu, bool tail_split, int stagger_ticks = 0) {
    const int BM = WM * FB * 16, BN = WN * FA * 16, NW = WN * WM;
    const int TM = (sh.M + BM - 1) / BM, TN = (sh.N + BN - 1) / BN, tiles = sh.batch * TM * TN;
    const int Mpad = TM * BM, Npad = TN * BN;
    const size_t lds_bytes = (size_t)NS * (WN * FA + WM * FB) * 3 * 1024;
    auto kern = k_gemm_b9v2<FA, FB, WN, WM, NS, SCHED>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * NW, lds_bytes));
    const int slots = ncu * (occ > 0 ? occ : 1), nkb = sh.K / 32;
    int n_full = tiles, rem = 0, split = 1;
    if (tail_split && tiles % slots) {
        n_full = (tiles / slots) * slots; rem = tiles - n_full;
        double best = 1e9;
        const int cand[] = {1, 2, 3, 4, 6, 8, 12};
        for (int c : cand) if (nkb / c >= 4) { const double t = (double)((rem * c + slots - 1) / slots) / c; if (t < best - 1e-9) { best = t; split = c; } }
        if (split == 1) { n_full = tiles; rem = 0; }
    }
    const long wStride = (long)(Npad / 16) * nkb * 192, xStride = (long)(Mpad / 16) * nkb * 192;
    static unsigned long long* dclk = nullptr;
    static f32x4* part = nullptr;
    if (!dclk) { CK(hipMalloc(&dclk, 48 * 65536)); CK(hipMemset(dclk, 0, 48 * 65536)); CK(hipMalloc(&part, (size_t)256 << 20)); }
    for (int b = 0; b < sh.batch; ++b) {
        const long nb = (long)(Npad / 16) * nkb, mb = (long)(Mpad / 16) * nkb;
        hipLaunchKernelGGL(k_split3, dim3((unsigned)((nb * 64 + 255) / 256)), dim3(256), 0, 0, dW + (size_t)b * sh.N * sh.K, wP + b * wStride, sh.N, sh.K, Npad);
        hipLaunchKernelGGL(k_split3, dim3((unsigned)((mb * 64 + 255) / 256)), dim3(256), 0, 0, dX + (size_t)b * sh.M * sh.K, xP + b * xStride, sh.M, sh.K, Mpad);
    }
    const int nwg = n_full + rem * split;
    auto gemm = [&]() {
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * NW), lds_bytes, 0, wP, xP, dOut, sh.M, sh.N, sh.K, wStride, xStride, TM, TN, n_full, split, part, dclk,
                           stagger_ticks);
        if (rem) hipLaunchKernelGGL((k_b9_fixup<FA, FB, WN, WM>), dim3(rem * NW), dim3(64), 0, 0, part, dOut, sh.M, sh.N, TM, TN, n_full, split);
    };
    CK(hipMemset(dOut, 0, (size_t)sh.batch * sh.M * sh.N * 4));
    gemm(); gemm();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {
        CK(hipEventRecord(e0)); gemm(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float one = 0; CK(hipEventElapsedTime(&one, e0, e1));
        const int nwarm = (int)(600.0f / (one > 0.01f ? one : 0.01f));
        for (int r = 0; r < nwarm; ++r) gemm();
        CK(hipDeviceSynchronize());
    }
    const int reps = 200;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) gemm();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms_g = 0;
    CK(hipEventElapsedTime(&ms_g, e0, e1));
    const double us_g = ms_g * 1e3 / reps, flop = 2.0 * sh.batch * sh.M * (double)sh.N * sh.K;
    std::vector<unsigned long long> hc(2 * (size_t)n_full);
    CK(hipMemcpy(hc.data(), dclk, 16 * (size_t)n_full, hipMemcpyDeviceToHost));
    // mixed load: the GEMM between HBM-bound kernels (3 x 0.25 ms of streaming per GEMM), as inside a batch of the pipeline
    double us_mix = 0, ghz_mix = 0;
    if (getenv("UB_MIX")) {
        static f32x4* big = nullptr;
        const size_t nbig = (size_t)96 << 20;                      // 1.5 GiB
        if (!big) { CK(hipMalloc(&big, nbig * 16)); CK(hipMemset(big, 0, nbig * 16)); }
        const int nmix = 150;
        std::vector<hipEvent_t> ev(2 * nmix);
        for (auto& e : ev) CK(hipEventCreate(&e));
        for (int r = 0; r < nmix + 100; ++r) {
            for (int q = 0; q < 3; ++q) hipLaunchKernelGGL(k_stream, dim3(ncu * 8), dim3(256), 0, 0, big, (float*)dOut, nbig);
            if (r >= 100) CK(hipEventRecord(ev[2 * (r - 100)]));
            gemm();
            if (r >= 100) CK(hipEventRecord(ev[2 * (r - 100) + 1]));
        }
        CK(hipDeviceSynchronize());
        for (int r = 0; r < nmix; ++r) { float ms = 0; CK(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1])); us_mix += ms * 1e3 / nmix; }
        std::vector<unsigned long long> hm(2 * (size_t)n_full);
        CK(hipMemcpy(hm.data(), dclk, 16 * (size_t)n_full, hipMemcpyDeviceToHost));
        for (int b = 0; b < n_full; ++b) ghz_mix += (double)hm[2 * b] / (double)hm[2 * b + 1] * 0.1 / n_full;
        for (auto& e : ev) CK(hipEventDestroy(e));
        gemm();
        CK(hipDeviceSynchronize());
    }
    std::vector<float> hOut((size_t)sh.batch * sh.M * sh.N);
    CK(hipMemcpy(hOut.data(), dOut, hOut.size() * 4, hipMemcpyDeviceToHost));
    double maxref = 0, maxerr = 0;
    const int bs[2] = {0, sh.batch - 1};
    for (int bi = 0; bi < (sh.batch > 1 ? 2 : 1); ++bi)
        for (int mi = 0; mi < 24; ++mi) {
            const int b = bs[bi], m = mi < 4 ? sh.M - 1 - mi : (int)(((long)mi * 2654435761u) % sh.M);     // incl. rows of the last (tail) tiles
            for (int n = 0; n < sh.N; n += 3) {
                double a = 0;
                const float* xr = &hX[((size_t)b * sh.M + m) * sh.K];
                const float* wr = &hW[((size_t)b * sh.N + n) * sh.K];
                for (int k = 0; k < sh.K; ++k) a += (double)xr[k] * (double)wr[k];
                maxref = fmax(maxref, fabs(a)); maxerr = fmax(maxerr, fabs((double)hOut[((size_t)b * sh.M + m) * sh.N + n] - a));
            }
        }
    double cyc = 0, ghz = 0;
    for (int b = 0; b < n_full; ++b) { cyc += (double)hc[2 * b]; ghz += (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1; }
    cyc /= n_full; ghz /= n_full;
    if (getenv("UB_LIFE")) {
        std::vector<unsigned long long> hl(2 * (size_t)n_full);
        CK(hipMemcpy(hl.data(), dclk + 3 * 65536, 16 * (size_t)n_full, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        double life = 0, loop = 0;
        for (int b = 0; b < n_full; ++b) { t0 = hl[2 * b] < t0 ? hl[2 * b] : t0; t1 = hl[2 * b + 1] > t1 ? hl[2 * b + 1] : t1; life += (double)(hl[2 * b + 1] - hl[2 * b]) * 0.01; loop += (double)hc[2 * b + 1] * 0.01; }
        printf("      workgroup lifetimes (last launch): first entry -> last exit %.1f us; mean lifetime %.1f us of which main loop %.1f us; sum of lifetimes / %d slots = %.1f us\n",
               (double)(t1 - t0) * 0.01, life / n_full, loop / n_full, slots, life / slots);
        // start-time histogram: how many workgroups entered in each 10 us window
        int hist[32] = {0};
        for (int b = 0; b < n_full; ++b) { const int k = (int)((hl[2 * b] - t0) / 1000); hist[k < 31 ? k : 31]++; }
        printf("      entries per 10 us window:");
        for (int k = 0; k < 20; ++k) printf(" %d", hist[k]);
        printf("\n");
    }
    const double ideal = 9.0 * FA * FB * 16.0 * occ * (NW / 4);
    if (stagger_ticks > 0) printf("  [second residency wave delayed by %.1f us]", stagger_ticks * 0.01);
    if (stagger_ticks < 0) printf("  [wave priorities, scheme %d]", -stagger_ticks);
    printf("  v2 tile %3dx%-3d %d waves NS=%d sched %d  %d wg/CU  %5d tiles = %d whole + %d x %d slabs | gemm %7.1f us = %6.1f TF-eq | vs library %.0f us: %.2fx"
           " | %.0f cyc/step (ratio %.2f) %.2f GHz | err %.2e\n",
           BN, BM, NW, NS, SCHED, occ, tiles, n_full, rem, split, us_g, flop / us_g * 1e-6, sh.yard_us, sh.yard_us / us_g, cyc / nkb, cyc / nkb / ideal, ghz,
           maxerr / maxref);
    if (us_mix > 0) printf("      between HBM-bound kernels (75 %% of the time streaming): gemm %7.1f us = %.2fx the library's, in-kernel clock %.2f GHz\n", us_mix, sh.yard_us / us_mix, ghz_mix);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int ncu = 256;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    float* dout;
    unsigned long long* dclk;
    CK(hipMalloc(&dout, 4 << 20));
    CK(hipMalloc(&dclk, 16 * 1024));
    std::vector<unsigned long long> hclk(2 * ncu);
    printf("== A. bare MFMA loops, 4-wave workgroups, one per CU (one wave per SIMD), random operands\n");
    for (int which = 0; which < 2; ++which) {
        const int iters = which == 0 ? 6000 : 6000;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < 3; ++r) {
            if (which == 0) hipLaunchKernelGGL(k_bare_f32, dim3(ncu), dim3(256), 0, 0, dout, dclk, iters, 77u);
            else hipLaunchKernelGGL(k_bare_b9, dim3(ncu), dim3(256), 0, 0, dout, dclk, iters, 77u);
        }
        CK(hipDeviceSynchronize());
        const int reps = 40;   // ~1 s of back-to-back launches so the clock settles
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            if (which == 0) hipLaunchKernelGGL(k_bare_f32, dim3(ncu), dim3(256), 0, 0, dout, dclk, iters, 77u);
            else hipLaunchKernelGGL(k_bare_b9, dim3(ncu), dim3(256), 0, 0, dout, dclk, iters, 77u);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hclk.data(), dclk, 16 * ncu, hipMemcpyDeviceToHost));
        double ghz = 0;
        for (int b = 0; b < ncu; ++b) ghz += (double)hclk[2 * b] / (double)hclk[2 * b + 1] * 0.1;
        ghz /= ncu;
        // f32: 8*16 MFMAs x 2048 flop per wave-iteration; b9: 144 MFMAs cover 16 fragment pairs x K = 32: 16 x 16384 flop-equivalent
        const double flop = which == 0 ? (double)ncu * 4 * iters * 128.0 * 2048.0 * reps : (double)ncu * 4 * iters * 16.0 * 16384.0 * reps;
        const double cyc_per_mfma = (double)hclk[0] / ((double)iters * (which == 0 ? 128 : 144));
        printf("  %-44s %7.1f TFLOP/s%s  in-kernel clock %.2f GHz  %.1f cycles per MFMA\n",
               which == 0 ? "v_mfma_f32_16x16x4_f32 chain:" : "9 x v_mfma_f32_16x16x32_bf16 (exact-f32 equiv.):", flop / (ms * 1e-3) / 1e12,
               which == 0 ? "   " : "-eq", ghz, cyc_per_mfma);
    }

    const Shape shapes[] = {{"ViT qkv  M 6272, K 768, N 2304", 1, 6272, 768, 2304, 192.0},
                            {"ViT fc2  M 6272, K 3072, N 768", 1, 6272, 3072, 768, 225.0},
                            {"cond blk 5 x (M 640, K 4096, N 4096)", 5, 640, 4096, 4096, 805.0}};
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    for (int si = 0; si < 3; ++si) {
        if (only >= 0 && si != only) continue;
        const Shape& sh = shapes[si];
        printf("== B. %s\n", sh.name);
        std::vector<float> hW((size_t)sh.batch * sh.N * sh.K), hX((size_t)sh.batch * sh.M * sh.K);
        for (auto& v : hW) v = frand() * 0.05f;
        for (auto& v : hX) v = frand();
        float *dW, *dX, *dOut;
        bf16x8 *wP, *xP;
        CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dOut, (size_t)sh.batch * sh.M * sh.N * 4));
        const size_t padW = (size_t)sh.batch * (sh.N + 256) * sh.K * 6, padX = (size_t)sh.batch * (sh.M + 256) * sh.K * 6;
        CK(hipMalloc(&wP, padW)); CK(hipMalloc(&xP, padX));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
        if (argc > 2) {
            run_gemm2<4, 2, 2, 2, 2, 0>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu, false);
            if (getenv("UB_STAGGER")) {
                for (int tk : {500, 1000, 1500, 2000}) run_gemm2<4, 2, 2, 2, 2, 0>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu, false, tk);
            }
            if (getenv("UB_PRIO")) {
                for (int sc : {-1, -2, -3, -4}) run_gemm2<4, 2, 2, 2, 2, 0>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu, false, sc);
                run_gemm2<4, 2, 2, 2, 2, 0>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu, false);      // the unprioritised form again, same clock state
                CK(hipFree(dW)); CK(hipFree(dX)); CK(hipFree(dOut)); CK(hipFree(wP)); CK(hipFree(xP));
                continue;
            }
            run_gemm2<4, 2, 2, 2, 2, 0>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu, true);
            run_gemm2<4, 2, 2, 4, 2, 0>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu, true);
            run_gemm2<4, 2, 2, 4, 3, 0>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu, true);
            run_gemm2<4, 4, 2, 2, 3, 0>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu, true);
            CK(hipFree(dW)); CK(hipFree(dX)); CK(hipFree(dOut)); CK(hipFree(wP)); CK(hipFree(xP));
            continue;
        }
        run_gemm<2, 2, 2, 9>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        run_gemm<4, 4, 2, 9>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        run_gemm<4, 4, 3, 9>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        run_gemm<4, 2, 2, 9>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        run_gemm<4, 2, 3, 9>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        run_gemm<2, 4, 2, 9>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        run_gemm<2, 2, 3, 9>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        run_gemm<4, 4, 2, 6>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        run_gemm<4, 2, 2, 6>(sh, dW, dX, wP, xP, dOut, hW, hX, ncu);
        CK(hipFree(dW)); CK(hipFree(dX)); CK(hipFree(dOut)); CK(hipFree(wP)); CK(hipFree(xP));
    }
    return 0;
}
