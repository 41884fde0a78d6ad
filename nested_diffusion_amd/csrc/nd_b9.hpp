// nd_b9.hpp -- exact-f32 GEMM core on the bf16 matrix pipe ("bf16 x 9", gfx950 / CDNA4 only).
//
// v_mfma_f32_16x16x4_f32 -- the instruction every fp32 GEMM of this library was built on -- runs at 1/16 of the bf16 MFMA rate.
// An fp32 value splits EXACTLY into three bf16 pieces
//     a1 = rn_bf16(a),  a2 = rn_bf16(a - a1),  a3 = a - a1 - a2          (8 + 8 + 8 significand bits; both subtractions exact)
// so a*b = sum over the nine pairs a_p*b_q, each pair product exact in fp32 (8 x 8 bits), accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16: 9 MFMAs of 16 cycles per (16 x 16 output fragment, K = 32) against 8 MFMAs of 32 cycles for the f32
// instruction -- 9/16 of the matrix-pipe cycles, the same arithmetic as before in a different summation order (measured error
// against fp64, tools/ubench_bf16x9.hip: 7.9e-7 of the largest output at K = 768, 1.5e-6 at K = 3072 -- the f32 kernels' figures).
// The chip holds a lower clock under bf16 MFMA load (1.75-1.95 GHz against ~2.3), so 9/16 of the cycles is 0.7-0.8 of the time.
//
// Operand layout "frag32b3": a K-contiguous matrix A[R][K] (R padded to 16, K % 32 == 0) is stored as blocks of 16 rows x 32 k;
// block (r/16, k/32) at 16-byte-unit offset ((r/16)*(K/32) + k/32)*192 holds three PLANES (a1, a2, a3) of 64 x 16 bytes: lane l of a
// plane = the 8 bf16 of row r%16 = l&15, k%32 = 8*(l>>4)..+7 -- the A/B operand of v_mfma_f32_16x16x32_bf16.  A plane goes
// global -> LDS with ONE global_load_lds_dwordx4 per wave and LDS -> registers with one conflict-free ds_read_b128 per lane.
// Operands are split ONCE by their producer (weights at load; activations in the epilogue that writes them), never inside the GEMM.
//
// Main loop (b9_mainloop): workgroup tile (WN*FA*16 rows of w) x (WM*FB*16 rows of x), WN x WM waves of FA x FB fragments; K-step
// 32; LDS ring of NS slots, two register sets, ONE barrier per step:
//     wait own LDS-DMA of step j+1 and own fragment reads of step j -> barrier -> 9*FA*FB MFMAs of step j with, dealt evenly between
//     them (sched_group_barrier), the LDS-DMA pieces of step j+NS into the slot the barrier released and the fragment reads of step
//     j+1 into the other register set.
// Measured (tools/ubench_bf16x9.hip, tools/ubench_ldsdma.hip): an LDS-DMA instruction occupies its wave's vector-memory path for 64
// cycles but costs the same wave's MFMA stream ~1 cycle; in cycles the loop runs at 1.00-1.08 of its MFMAs alone.
#pragma once
#include "nd_common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define B9_BLOCK_UNITS 192          // 16-byte units per (16 rows x 32 k) block: 3 planes x 64 lanes

// bytes of the frag32b3 image of an [R][K] matrix
static inline size_t nd_b9_bytes(int R, int K) { return (size_t)((R + 15) / 16) * (K / 32) * B9_BLOCK_UNITS * 16; }

// the three bf16 pieces of one fp32 value (exact: a == (float)h1 + (float)h2 + (float)h3 for every finite a whose pieces do not
// underflow, i.e. |a| > 2^-110 or a == 0).  An infinite a (or one that rounds to an infinite bf16) keeps h1 = +-inf and zero low
// pieces, so that it multiplies like an infinity (inf - inf would make the low pieces NaN); a NaN stays a NaN.
__device__ __forceinline__ void nd_b9_split(float a, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)a;
    const float f1 = (float)h1;
    const float r1 = __builtin_isinf(f1) ? 0.f : a - f1;      // (A/B on one box: the guard costs nothing measurable in the conditioner)
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);
}

// Store 4 consecutive k (k % 4 == 0) of row r of an [R][K] activation matrix into its frag32b3 image: three 8-byte pieces.
// `img` is the image base; nkb = K / 32.
__device__ __forceinline__ void nd_b9_store4(bf16x8* img, int nkb, int r, int k, float v0, float v1, float v2, float v3) {
    bf16x4 p1, p2, p3;
    __bf16 a, b, c;
    nd_b9_split(v0, a, b, c); p1[0] = a; p2[0] = b; p3[0] = c;
    nd_b9_split(v1, a, b, c); p1[1] = a; p2[1] = b; p3[1] = c;
    nd_b9_split(v2, a, b, c); p1[2] = a; p2[2] = b; p3[2] = c;
    nd_b9_split(v3, a, b, c); p1[3] = a; p2[3] = b; p3[3] = c;
    const size_t blk = (size_t)(r >> 4) * nkb + (k >> 5);
    const int lane = (r & 15) + 16 * ((k & 31) >> 3), half = (k & 7) >> 2;
    bf16x4* q = reinterpret_cast<bf16x4*>(img + blk * B9_BLOCK_UNITS + lane) + half;
    *(__attribute__((address_space(1))) bf16x4*)(q) = p1;
    *(__attribute__((address_space(1))) bf16x4*)(q + 2 * 64) = p2;
    *(__attribute__((address_space(1))) bf16x4*)(q + 4 * 64) = p3;
}

// ---- "qkv images": the attention operands of one (image, head), written by the qkv GEMM's epilogue (nd_gemm_b9.hip, ATT form) and
// read by k_attention_b9 (nd_attention.hip).  One record per (image b, head h) at 16-byte-unit offset (b*heads + h) * units():
//   Q   nfq x 2 blocks   block (f, c) = (16 query rows of fragment f) x (d = 32c .. 32c+31): plain frag32b3 blocks, k = d
//   K   nfq x 2 blocks   the same for the keys
//   V^T 4 x nkb blocks   block (df, kb) = (d = 16 df .. +15) x (32 keys of block kb), k = key, the keys of a block PERMUTED: position
//                        (lane group g, element e) of a plane holds key 32 kb + 16 (e >> 2) + 4 g + (e & 3) -- the two runs of four
//                        keys whose scores a lane of the attention kernel already holds in score fragments 2 kb and 2 kb + 1
//                        (S[key = 16 f + 4 g + r][q]), so the probabilities feed the P V MFMA without any cross-lane movement.
// nfq = ceil(ntok / 16), nkb = ceil(ntok / 32).  Rows / keys past ntok are never written (whatever the buffer held stays there): the
// attention kernel masks them (scores by select, V operands by zeroing) and never stores their outputs.
struct B9AttLayout {
    int ntok, heads;
    __host__ __device__ int nfq() const { return (ntok + 15) >> 4; }
    __host__ __device__ int nkb() const { return (ntok + 31) >> 5; }
    __host__ __device__ int q_block0() const { return 0; }
    __host__ __device__ int k_block0() const { return 2 * nfq(); }
    __host__ __device__ int v_block0() const { return 4 * nfq(); }
    __host__ __device__ size_t units() const { return (size_t)(4 * nfq() + 4 * nkb()) * B9_BLOCK_UNITS; }
};

// 4 values as three 8-byte pieces into (block, lane, half) of an image
__device__ __forceinline__ void nd_b9_store4_at(bf16x8* blockp, int lane, int half, float v0, float v1, float v2, float v3) {
    bf16x4 p1, p2, p3;
    __bf16 a, b, c;
    nd_b9_split(v0, a, b, c); p1[0] = a; p2[0] = b; p3[0] = c;
    nd_b9_split(v1, a, b, c); p1[1] = a; p2[1] = b; p3[1] = c;
    nd_b9_split(v2, a, b, c); p1[2] = a; p2[2] = b; p3[2] = c;
    nd_b9_split(v3, a, b, c); p1[3] = a; p2[3] = b; p3[3] = c;
    bf16x4* q = reinterpret_cast<bf16x4*>(blockp + lane) + half;
    *(__attribute__((address_space(1))) bf16x4*)(q) = p1;
    *(__attribute__((address_space(1))) bf16x4*)(q + 2 * 64) = p2;
    *(__attribute__((address_space(1))) bf16x4*)(q + 4 * 64) = p3;
}

// the NP LDS-DMA pieces of one wave and K-step; bit pc of NTMASK: piece pc is requested with the nontemporal policy
template <int PC, int NP, unsigned NTMASK>
__device__ __forceinline__ void b9_stage_pieces(const bf16x8* const (&src)[NP], size_t step_units, bf16x8* slot, int first, int last) {
    if constexpr (PC < NP) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[PC] + step_units),
                                         (__attribute__((address_space(3))) void*)&slot[min(first + PC, last) * 64], 16, 0,
                                         ((NTMASK >> PC) & 1u) ? 2 : 0);
        b9_stage_pieces<PC + 1, NP, NTMASK>(src, step_units, slot, first, last);
    }
}

// One workgroup's K loop.  src[u]: this wave's NP LDS-DMA sources at K-step 0 of its range (per-lane pointers: piece base + lane),
// advanced by B9_BLOCK_UNITS per step; piece e = min(wave*NP + u, NPC-1) of a slot = (fragment e/3, plane e%3), fragments
// 0 .. WN*FA-1 = the tile's w fragments, the rest its x fragments.  acc[i][j] += w fragment (wn*FA + i) x x fragment (wm*FB + j):
// lane l of acc[i][j] holds D[n = 4*(l>>4) + r][m = l&15], i.e. 4 consecutive output columns n of activation row m.
// SWAP: the operand roles of the MFMA exchanged (A = x fragment, B = w fragment): lane l of acc[i][j] then holds D[m = 4*(l>>4) + r][n = l&15],
// i.e. 4 consecutive ACTIVATION rows m of output column n -- the form whose frag32b3 store is the TRANSPOSED image (k = m contiguous):
// the V^T operand of the attention's P V contraction (nd_attention.hip, k_attention_b9).  Same products, same order of terms.
template <int FA, int FB, int WN, int WM, int NS, unsigned NTMASK = 0u, bool SWAP = false>
__device__ __forceinline__ void b9_mainloop_impl(f32x4 (&acc)[FA][FB], const bf16x8* (&src)[(((WN * FA + WM * FB) * 3) + WN * WM - 1) / (WN * WM)],
                                                 bf16x8* lds, int nk, int wave, int wn, int wm, int lane) {
    constexpr int NW = WN * WM, NFRAG = WN * FA + WM * FB, NPC = NFRAG * 3, NP = (NPC + NW - 1) / NW;
    constexpr int NM = 9 * FA * FB, NRD = 3 * (FA + FB), NMEM = NP + NRD, RATIO = NM / NMEM > 0 ? NM / NMEM : 1;
    static_assert(NS >= 2 && NP * (NS - 1) < 64, "vmcnt is a 6-bit counter");
#define B9_STAGE(slot, step) b9_stage_pieces<0, NP, NTMASK>(src, (size_t)(step) * B9_BLOCK_UNITS, lds + (size_t)(slot) * NPC * 64, wave * NP, NPC - 1);
#define B9_READ(set, slot)                                                                                                          \
    {                                                                                                                               \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                             \
            _Pragma("unroll") for (int i = 0; i < FA; ++i) fw[set][p][i] = lds[(((slot) * NFRAG + wn * FA + i) * 3 + p) * 64 + lane]; \
            _Pragma("unroll") for (int j = 0; j < FB; ++j) fx[set][p][j] = lds[(((slot) * NFRAG + WN * FA + wm * FB + j) * 3 + p) * 64 + lane]; \
        }                                                                                                                           \
    }
#define B9_TERM(set, p, q)                                                                                                          \
    {                                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < FA; ++i)                                                                              \
            _Pragma("unroll") for (int j = 0; j < FB; ++j)                                                                          \
                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[set][q][j], fw[set][p][i], acc[i][j], 0, 0, 0)        \
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[set][p][i], fx[set][q][j], acc[i][j], 0, 0, 0);       \
    }
    bf16x8 fw[2][3][FA], fx[2][3][FB];
    // prologue: steps 0 .. NS-1 -> slots 0 .. NS-1 (clamped past the end: valid data nobody reads), step 0 into register set 0
#pragma unroll
    for (int u = 0; u < NS; ++u) B9_STAGE(u, min(u, nk - 1))
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    B9_READ(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int U = (NS % 2 == 0) ? NS : 2 * NS;     // unroll: slot and register-set indices are compile-time constants
    for (int s = 0; s < nk; s += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s + u < nk) {
                // all but the NS-2 youngest steps' LDS-DMA of this wave landed + own fragment reads done, then everybody's
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP * (NS - 2)) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                B9_STAGE(u % NS, min(s + u + NS, nk - 1))
                B9_READ((u + 1) & 1, (u + 1) % NS)
                // smallest pair products first (a3 b3 ... a1 b1): the order is fixed, results are reproducible
                B9_TERM(u & 1, 2, 2) B9_TERM(u & 1, 2, 1) B9_TERM(u & 1, 1, 2)
                B9_TERM(u & 1, 2, 0) B9_TERM(u & 1, 0, 2) B9_TERM(u & 1, 1, 1) B9_TERM(u & 1, 1, 0) B9_TERM(u & 1, 0, 1) B9_TERM(u & 1, 0, 0)
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
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's use of its LDS
#undef B9_STAGE
#undef B9_READ
#undef B9_TERM
}

// NTW: the w pieces (fragments 0 .. WN*FA-1 of a slot) are requested nontemporally -- for launches whose w panels are read once while
// their x panels are re-read by every column tile, so that the stream of w does not push x out of the Infinity Cache.  Which of a
// wave's pieces are w pieces depends on the wave: the loop is instantiated per count of leading w pieces (wave-uniform switch).
template <int FA, int FB, int WN, int WM, int NS, bool NTW = false, bool SWAP = false>
__device__ __forceinline__ void b9_mainloop(f32x4 (&acc)[FA][FB], const bf16x8* (&src)[(((WN * FA + WM * FB) * 3) + WN * WM - 1) / (WN * WM)],
                                            bf16x8* lds, int nk, int wave, int wn, int wm, int lane) {
    constexpr int NW = WN * WM, NPC = (WN * FA + WM * FB) * 3, NP = (NPC + NW - 1) / NW;
    if constexpr (!NTW) {
        b9_mainloop_impl<FA, FB, WN, WM, NS, 0u, SWAP>(acc, src, lds, nk, wave, wn, wm, lane);
    } else {
        static_assert(!SWAP, "the transposed-output form has no nontemporal variant");
        static_assert(NP <= 6, "one instantiation per count of w pieces");
        const int nw = min(max(WN * FA * 3 - wave * NP, 0), NP);
        switch (nw) {
            case 0: b9_mainloop_impl<FA, FB, WN, WM, NS, 0u>(acc, src, lds, nk, wave, wn, wm, lane); break;
            case 1: b9_mainloop_impl<FA, FB, WN, WM, NS, 1u>(acc, src, lds, nk, wave, wn, wm, lane); break;
            case 2: b9_mainloop_impl<FA, FB, WN, WM, NS, 3u>(acc, src, lds, nk, wave, wn, wm, lane); break;
            case 3: b9_mainloop_impl<FA, FB, WN, WM, NS, 7u>(acc, src, lds, nk, wave, wn, wm, lane); break;
            case 4: b9_mainloop_impl<FA, FB, WN, WM, NS, 15u>(acc, src, lds, nk, wave, wn, wm, lane); break;
            case 5: b9_mainloop_impl<FA, FB, WN, WM, NS, 31u>(acc, src, lds, nk, wave, wn, wm, lane); break;
            default: b9_mainloop_impl<FA, FB, WN, WM, NS, 63u>(acc, src, lds, nk, wave, wn, wm, lane); break;
        }
    }
}
