// nd_attention.hip -- the fp32 attention core of the ViT blocks (timm 0.4.12 Attention.forward, head dim 64; call sites
// classification_train_separately.py:337-340).  gfx950 only.  Built with -mllvm -amdgpu-mfma-vgpr-form
// (nested_diffusion_amd/build.py): the score accumulators are consumed by the softmax (VALU) and fed back as MFMA operands, so
// keeping them in AGPRs costs a v_accvgpr move per value each way.
#include "nd_common.hpp"
#include "nd_b9.hpp"

// RING form (the default fp32 kernel).  A workgroup is FOUR waves -- one per SIMD -- owning up to four 16-row query fragments
// of one (image, head): ceil(NF/4) workgroups per head (NF = 13 at N = 196: 4 + 3 + 3 + 3 fragments), 1536 workgroups at
// B = 32 x 12 heads.  K and then V pass through a two-slot LDS ring as 2 NT tiles K0 .. K(NT-1), V0 .. V(NT-1) of TF = ceil(NF/NT)
// key fragments (NT = 3, TF = 5 at NF = 13: 40 KiB of LDS, four workgroups resident per CU: 160 KiB exactly, and 110 VGPRs allow 4 waves per SIMD), brought by LDS-DMA
// (global_load_lds_dwordx4: one 1 KiB piece = 4 key rows per wave instruction, no VGPR round trip).  Tile t + 2 is requested as
// soon as tile t has been consumed, so every tile but the first has a whole compute phase to land:
//     DMA K0, K1 | S += K0 q | DMA K2 | S += K1 q | DMA V0 | S += K2 q | DMA V1 | softmax | O += P V0 | DMA V2 | O += P V1 | O += P V2
// The score fragments of all keys stay in registers, the softmax is the exact two-pass form (max, exp, sum, normalise) of
// the kernels above: same arithmetic, same operand maps, bit-identical results.
// LDS images are unpadded 256-byte rows (an LDS-DMA piece is written lane-linear).  K rows are stored with their 16-byte
// chunks XOR-swizzled by (row & 15) -- applied on the DMA's per-lane SOURCE address and again on the read -- so the MFMA
// operand read (16 key rows x one chunk column per ds_read_b128 lane group) touches every bank once; V is read along rows
// (16 lanes = the 16 chunks of one row) and needs no swizzle.  Rows past N are clamped copies (finite; masked by the softmax).
// The DMA is issued from inline assembly: hipcc otherwise drains vmcnt before the first LDS read that follows ANY pending
// LDS-DMA (it cannot tell the slots apart), which would expose every tile's latency right after its request.  All ordering is
// explicit: a counted s_waitcnt vmcnt by the issuing wave + a barrier before a slot is read; lgkmcnt(0) + a barrier before it
// is refilled.
// one LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane byte offset) to LDS bytes [lds_byte_addr, +1024)
// m0 (the DMA's LDS base) is a reserved register the compiler also uses itself, so it must not appear in a clobber list
// ("may not be preserved across the asm statement"): the statement saves it into a compiler-allocated SGPR and restores it,
// i.e. m0 is unchanged across the statement by construction.  The DMA samples m0 when it issues, so restoring right behind it
// is the same sequence the compiler emits for back-to-back __builtin_amdgcn_global_load_lds with different bases.
__device__ __forceinline__ void nd_lds_dma16(const float* sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    unsigned saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "v"(voff_bytes), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// exp(x) for x <= 0 (softmax arguments), ~1 ulp: exp2 of the product x*log2(e) carried in two pieces (t rounded + its exact
// residual + the low part of log2 e), first-order correction on the result.  v_exp_f32 is the only transcendental; no range
// handling is needed below zero (underflow flushes to 0, as the softmax wants).  x must be finite.
__device__ __forceinline__ float nd_exp_neg(float x) {
    const float L2E = 1.44269504088896340736f, L2E_LO = 1.92596299e-8f, LN2 = 0.69314718055994530942f;
    const float t = x * L2E;
    const float r = __builtin_fmaf(x, L2E, -t) + x * L2E_LO;
    const float p = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(p, r * LN2, p);
}

#ifdef ND_ATT_STAMPS
__device__ long long* nd_att_stamps = nullptr;     // tools/att_stamps.py (debug build only): 16 clocks per wave
extern "C" int nd_debug_set_att_stamps(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(nd_att_stamps), &p, sizeof p) == hipSuccess ? 0 : -1; }
#define ATT_STAMP(i) { if (nd_att_stamps && lane == 0) nd_att_stamps[((size_t)blockIdx.x * 4 + wave) * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); }
#else
#define ATT_STAMP(i)
#endif
template <int NF, int NT>
__global__ __launch_bounds__(256) void k_attention_ring(const float* __restrict__ qkv, float* __restrict__ out, int B, int N, int heads,
                                                        int QG, int split_out) {
    constexpr int TF = (NF + NT - 1) / NT;                // key fragments per tile (the last tile may hold fewer)
    constexpr int SLOT = TF * 16 * 64;                    // floats per slot
    __shared__ __attribute__((aligned(16))) float smem[2 * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups b and b + 8 share an XCD: each XCD takes a contiguous run of (head, query group) pairs, so the QG
    // workgroups of a head read its K / V through one L2
    const int total = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = total / 8, r = total % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bh = bid / QG, qg = bid - bh * QG;
    const int b = bh / heads, hd = bh - b * heads;
    const int base_n = NF / QG, rem_n = NF - base_n * QG;
    const int nq = base_n + (qg < rem_n ? 1 : 0);                 // query fragments of this workgroup (<= 4)
    const int qf = qg * base_n + min(qg, rem_n) + wave;           // this wave's query fragment
    const bool active = wave < nq;
    const int Cm = heads * 64;
    const size_t rs = (size_t)3 * Cm;
    const float* base = qkv + (size_t)b * N * rs + (size_t)hd * 64;
    const float* qb = base;
    const float* kb = base + Cm;
    const float* vb = base + 2 * Cm;
    const int g = lane >> 4, li = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;
    ATT_STAMP(0)

    // tile T (0 .. 2NT-1; K tiles first) -> slot T & 1: pieces of 4 rows (1 KiB), piece p = wave + 4 i, so every wave issues
    // exactly nf(T) LDS-DMAs per tile (what the counted waits below rely on); lane l of a piece: row l>>4, chunk l&15.
    // Row inside the tile = 16 i + (4 wave + g): the swizzle term (row & 15) and the lane's byte offset do not depend on i.
    const int r16 = 4 * wave + g;                                      // row & 15 of every row this lane fetches
    const unsigned offK = (unsigned)(((size_t)r16 * rs + 4 * (li ^ r16)) * sizeof(float));
    const unsigned offV = (unsigned)(((size_t)r16 * rs + 4 * li) * sizeof(float));
    auto stage = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        constexpr bool isK = T < NT;
        constexpr int F0 = (T % NT) * TF, NFT = (NF - F0 < TF ? NF - F0 : TF);
        const float* src = isK ? kb : vb;
#pragma unroll
        for (int i = 0; i < NFT; ++i) {
            const int row0 = 16 * (F0 + i);                            // first row of this 16-row fragment (uniform)
            const unsigned dst = lds0 + (unsigned)(((T & 1) * SLOT + (wave + 4 * i) * 256) * sizeof(float));
            if (row0 + 16 <= N) {
                nd_lds_dma16(src + (size_t)row0 * rs, isK ? offK : offV, dst);
            } else {                                                   // last fragment (row0 < N <= row0 + 15): rows clamped per lane
                const int row = min(row0 + r16, N - 1);
                const unsigned off = (unsigned)(((size_t)(row - row0) * rs + 4 * (isK ? (li ^ r16) : li)) * sizeof(float));
                nd_lds_dma16(src + (size_t)row0 * rs, off, dst);
            }
        }
    };
    stage(std::integral_constant<int, 0>{});
    const int qrow = min(qf * 16 + li, N - 1);
    float4 qv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qv[c] = nd_ld16<false>(qb + (size_t)qrow * rs + 16 * c + 4 * g);
    stage(std::integral_constant<int, 1>{});
    f32x4 s[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) s[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    float inv = 0.f;

    // S fragments of a K tile, d-chunk c outermost: the NFT fragments' chunk-c operands are read while chunk c-1 is multiplied,
    // and consecutive MFMAs go to different fragments' accumulators (a chain on one accumulator would pay the 40-cycle
    // dependent latency on every 32-cycle MFMA)
    auto scores = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        constexpr int F0 = T * TF, NFT = (NF - F0 < TF ? NF - F0 : TF);
        const float* sk = smem + (T & 1) * SLOT + li * 64;
        f32x4 kv[2][NFT];
#pragma unroll
        for (int f = 0; f < NFT; ++f) kv[0][f] = *reinterpret_cast<const f32x4*>(sk + 16 * f * 64 + 4 * (g ^ li));
        __builtin_amdgcn_sched_barrier(0);             // all of chunk 0's reads in flight before the first MFMA (one latency, not NFT)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float qq[4] = {qv[c].x, qv[c].y, qv[c].z, qv[c].w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                // chunk c+1's operand reads, dealt over the four k-steps of chunk c (fragment f goes with k-step f % 4)
                if (c + 1 < 4) {
#pragma unroll
                    for (int f = jj; f < NFT; f += 4)
                        kv[(c + 1) & 1][f] = *reinterpret_cast<const f32x4*>(sk + 16 * f * 64 + 4 * ((4 * (c + 1) + g) ^ li));
                }
#pragma unroll
                for (int f = 0; f < NFT; ++f) s[F0 + f] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[c & 1][f][jj], qq[jj], s[F0 + f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);     // pins the order: consecutive MFMAs on different accumulators, reads in their shadow
            }
        }
    };
    auto pv = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        constexpr int F0 = (T - NT) * TF, NFT = (NF - F0 < TF ? NF - F0 : TF);
        const float* sv = smem + (T & 1) * SLOT + 4 * g * 64 + 4 * li;
        f32x4 vv[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[0][r] = *reinterpret_cast<const f32x4*>(sv + r * 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NFT; ++f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (f + 1 < NFT) vv[(f + 1) & 1][r] = *reinterpret_cast<const f32x4*>(sv + (16 * (f + 1) + r) * 64);
                const float p = s[F0 + f][r] * inv;              // normalised first, as torch (softmax then @ v)
                o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[f & 1][r][0], p, o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[f & 1][r][1], p, o[1], 0, 0, 0);
                o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[f & 1][r][2], p, o[2], 0, 0, 0);
                o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[f & 1][r][3], p, o[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // softmax over the keys (per query column q = lane & 15): in-lane over (f, r), across the 4 lane groups by xor 16 / 32
    auto softmax = [&]() {
        const float scale = 0.125f;  // 64^-0.5
        float mx = -1.0e30f;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * f + 4 * g + r;
                const float v = key < N ? s[f][r] * scale : -1.0e30f;   // finite stand-in for -inf: exp -> exactly 0
                s[f][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = nd_exp_neg(s[f][r] - mx);
                s[f][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        inv = 1.0f / sum;
    };

    // the tile loop, unrolled at compile time: tile T lives in slot T & 1; before its reads, all but the pieces of tile T + 1
    // (this wave's, the youngest in its queue) must have landed
    auto step = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        constexpr int NEXT = T + 1 < 2 * NT ? (NF - ((T + 1) % NT) * TF < TF ? NF - ((T + 1) % NT) * TF : TF) : 0;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEXT) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ATT_STAMP(1 + 2 * T)
        if (active) {
            if constexpr (T < NT) scores(tc);
            else pv(tc);
        }
        ATT_STAMP(2 + 2 * T)
        if constexpr (T + 2 < 2 * NT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // every wave is done reading slot T & 1
            __builtin_amdgcn_sched_barrier(0);
            stage(std::integral_constant<int, T + 2>{});
        }
        if constexpr (T == NT - 1) softmax();                      // under the landing of the first V tiles
    };
    auto run = [&](auto... tcs) { (step(tcs), ...); };
    if constexpr (NT == 1) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    else if constexpr (NT == 2) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{},
                                    std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
             std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
    // o[e][r'] = O[q = l&15][d = 4*(4g + r') + e]  ->  float4 over e at d0 = 16g + 4r'
    const int qo = qf * 16 + li;
    if (active && qo < N) {
        if (split_out) {
            // `out` is the frag32b3 image of [B*N, heads*64] (csrc/nd_b9.hpp): the input form of the proj Linear that follows
#pragma unroll
            for (int r = 0; r < 4; ++r)
                nd_b9_store4(reinterpret_cast<bf16x8*>(out), Cm >> 5, b * N + qo, hd * 64 + 16 * g + 4 * r, o[0][r], o[1][r], o[2][r], o[3][r]);
        } else {
            float* op = out + ((size_t)b * N + qo) * Cm + (size_t)hd * 64 + 16 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(op + 4 * r) = make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
        }
    }
    ATT_STAMP(13)
}

template <int NF>
static hipError_t launch_attention_ring(const float* qkv, float* out, int B, int N, int heads, int split_out, hipStream_t st) {
    // tiles per operand: 3 where that leaves whole tiles to stream (NF >= 6), else 2 / 1
    constexpr int NT = NF >= 6 ? 3 : (NF >= 2 ? 2 : 1);
    const int QG = (NF + 3) / 4;
    hipLaunchKernelGGL((k_attention_ring<NF, NT>), dim3(B * heads * QG), dim3(256), 0, st, qkv, out, B, N, heads, QG, split_out);
    return hipGetLastError();
}

hipError_t nd_launch_attention_ring(const float* qkv, float* out, int B, int N, int heads, int split_out, hipStream_t st) {
    switch ((N + 15) / 16) {
#define AR_CASE(NFV) case NFV: return launch_attention_ring<NFV>(qkv, out, B, N, heads, split_out, st);
        AR_CASE(1) AR_CASE(2) AR_CASE(3) AR_CASE(4) AR_CASE(5) AR_CASE(6) AR_CASE(7) AR_CASE(8)
        AR_CASE(9) AR_CASE(10) AR_CASE(11) AR_CASE(12) AR_CASE(13) AR_CASE(14) AR_CASE(15) AR_CASE(16)
#undef AR_CASE
    }
    return hipErrorInvalidValue;
}
