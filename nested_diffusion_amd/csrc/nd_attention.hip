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


// =================================================================================================================================
// bf16 x 9 form (round 5): both contractions on the bf16 matrix pipe with exact fp32 products (csrc/nd_b9.hpp) -- the last MFMA kernel
// of the path that still ran on v_mfma_f32_16x16x4_f32 (1/16 of the bf16 rate).  Operands are the per-(image, head) "qkv images" the
// qkv Linear's epilogue writes (nd_gemm_split_qkv; layout: B9AttLayout in nd_b9.hpp): Q and K as plain frag32b3 blocks (k = d), V
// TRANSPOSED (k = key, the keys of a 32-block in the order the score accumulators hold them), so nothing is split or transposed here
// except the probabilities, which are split in registers after the fp32 softmax:
//     S[key][q]  = sum_d K[key][d] Q[q][d]         18 MFMAs per (16 keys x 16 queries) = 288 matrix-pipe cycles (f32 form: 512)
//     O^T[d][q]  = sum_key V^T[d][key] P[q][key]   9 MFMAs per (16 d x 16 queries x 32 keys)            (1008 against 1664 per q fragment)
// A workgroup is four waves (one per SIMD) owning up to EIGHT query fragments of one (image, head) -- two per wave, so every K / V
// operand read from LDS feeds two fragments' MFMAs (one fragment per wave would keep the LDS 2/3 busy with operand reads alone: 6
// ds_read_b128 per 18 MFMAs on each of four SIMDs) -- ceil(NF / 8) workgroups per head (7 + 6 fragments at N = 196).  K and then V^T
// pass through a two-slot LDS ring in tiles of 4 key fragments / 2 key blocks (24 KiB each; 48 KiB and <= 256 VGPRs: two workgroups per CU), brought by LDS-DMA exactly as in the
// ring kernel above (inline-asm issue, counted waits, two barriers per tile).  Softmax: the same exact two-pass form in fp32.
// Keys past N: their scores are replaced by select (whatever the never-written rows of the K image hold), their V^T columns are zeroed
// in registers (0 x NaN would poison the sum), query rows past N are never stored.
template <typename F, int... Ts>
__device__ __forceinline__ void run_steps(F& step, std::integer_sequence<int, Ts...>) { (step(std::integral_constant<int, Ts>{}), ...); }

__device__ __forceinline__ bf16x8 nd_ld_b8(const bf16x8* p) { return *(const __attribute__((address_space(1))) bf16x8*)p; }

template <int NF>
__global__ __launch_bounds__(256, 2) void k_attention_b9(const bf16x8* __restrict__ att, float* __restrict__ out, int B, int N, int heads, int QG,
                                                      int split_out) {
    constexpr int NKB = (NF + 1) / 2;
#ifndef ND_AB9_TFK
#define ND_AB9_TFK 4
#endif
#ifndef ND_AB9_TKV
#define ND_AB9_TKV 2
#endif
#ifndef ND_AB9_NS
#define ND_AB9_NS 2
#endif
#ifndef ND_AB9_ABL          // timing ablations (WRONG results; variant builds of tools/ only): 1 no softmax, 2 no P split, 4 no score MFMAs,
#define ND_AB9_ABL 0        // 8 no P V MFMAs, 16 operand reads only once per tile, 32 no LDS-DMA after tile 0
#endif
    constexpr int TFK = NF < ND_AB9_TFK ? NF : ND_AB9_TFK;     // key fragments per K tile (6 KiB each)
    constexpr int TKV = NKB < ND_AB9_TKV ? NKB : ND_AB9_TKV;   // 32-key blocks per V^T tile (4 d-fragments x 3 KiB each)
    constexpr int NTK = (NF + TFK - 1) / TFK, NTV = (NKB + TKV - 1) / TKV, NTT = NTK + NTV;
    constexpr int SLOT = (TFK * 6 > TKV * 12 ? TFK * 6 : TKV * 12);      // 1 KiB pieces per slot
    constexpr int NS = ND_AB9_NS;                              // ring slots
    extern __shared__ __attribute__((aligned(16))) bf16x8 smem[];        // [NS][SLOT][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        const int total = gridDim.x, q = total / 8, r = total % 8, xcd = bid % 8, loc = bid / 8;      // a head's workgroups share an XCD
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bh = bid / QG, qg = bid - bh * QG;
    const int b = bh / heads, hd = bh - b * heads;
    const int base_n = NF / QG, rem_n = NF - base_n * QG;
    const int nq = base_n + (qg < rem_n ? 1 : 0);                  // query fragments of this workgroup (<= 8)
    const int qf0 = qg * base_n + min(qg, rem_n);
    const int njw = wave + 4 < nq ? 2 : (wave < nq ? 1 : 0);       // fragments of this wave: qf0 + wave (+ 4)
    const int g = lane >> 4, li = lane & 15;
    const B9AttLayout al{N, heads};
    const bf16x8* rec = att + (size_t)bh * al.units();
    const bf16x8* qimg = rec;
    const bf16x8* kimg = rec + (size_t)al.k_block0() * B9_BLOCK_UNITS;
    const bf16x8* vimg = rec + (size_t)al.v_block0() * B9_BLOCK_UNITS;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) bf16x8*)smem;
    const unsigned voff = (unsigned)lane * 16u;

    // tile T (K tiles first) -> slot T & 1.  Every wave issues exactly NPW(T) pieces per tile (the last ones re-issue the tile's last
    // piece: same bytes to the same place), which is what the counted waits rely on.
    auto stage = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr ((ND_AB9_ABL & 32) && T > 0) return;
        if constexpr (T < NTK) {
            constexpr int F0 = T * TFK, NFT = (NF - F0 < TFK ? NF - F0 : TFK), NPC = NFT * 6, NPW = (NPC + 3) / 4;
#pragma unroll
            for (int u = 0; u < NPW; ++u) {
                const int pc = min(wave * NPW + u, NPC - 1);
                nd_lds_dma16(reinterpret_cast<const float*>(kimg + (size_t)(F0 * 6 + pc) * 64), voff, lds0 + (unsigned)(((T % NS) * SLOT + pc) * 1024));
            }
        } else {
            constexpr int KB0 = (T - NTK) * TKV, NKT = (NKB - KB0 < TKV ? NKB - KB0 : TKV), NPC = NKT * 12, NPW = (NPC + 3) / 4;
#pragma unroll
            for (int u = 0; u < NPW; ++u) {
                const int pc = min(wave * NPW + u, NPC - 1);
                const int df = pc / (3 * NKT), kl = (pc / 3) % NKT, pl = pc % 3;             // LDS order [df][kl][plane]
                nd_lds_dma16(reinterpret_cast<const float*>(vimg + (size_t)((df * NKB + KB0 + kl) * 3 + pl) * 64), voff,
                             lds0 + (unsigned)(((T % NS) * SLOT + pc) * 1024));
            }
        }
    };
    auto pieces_per_wave = [](int T) constexpr {
        if (T >= NTT || ((ND_AB9_ABL & 32) && T > 0)) return 0;
        if (T < NTK) { const int f0 = T * TFK, nft = NF - f0 < TFK ? NF - f0 : TFK; return (nft * 6 + 3) / 4; }
        const int kb0 = (T - NTK) * TKV, nkt = NKB - kb0 < TKV ? NKB - kb0 : TKV;
        return (nkt * 12 + 3) / 4;
    };

    auto body = [&](auto njc) {
        constexpr int NJ = decltype(njc)::value, NJA = NJ > 0 ? NJ : 1;
        // this wave's query operands straight from the image (B operand of the score MFMAs: rows = queries, k = d), requested FIRST, then
        // tile 0; both are waited for together, the registers are handed to the compiler as defined by an asm, and only then are the
        // other tiles of the ring requested: the compiler does not see the inline-asm LDS-DMAs in vmcnt, so its own wait before the
        // first use of a loaded register is vmcnt(0) -- which must not find tiles 1 .. NS-1 in flight.
        bf16x8 qr[NJA][2][3];
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            const int qfj = min(qf0 + wave + 4 * j, NF - 1);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) qr[j][c][pl] = nd_ld_b8(qimg + (size_t)((qfj * 2 + c) * 3 + pl) * 64 + lane);
        }
        stage(std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NJA; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(qr[j][c][pl]));
        if constexpr (NTT > 1) stage(std::integral_constant<int, 1>{});
        if constexpr (NS > 2 && NTT > 2) stage(std::integral_constant<int, 2>{});
        if constexpr (NS > 3 && NTT > 3) stage(std::integral_constant<int, 3>{});
        f32x4 s[NJA][NF];
#pragma unroll
        for (int j = 0; j < NJA; ++j)
#pragma unroll
            for (int f = 0; f < NF; ++f) s[j][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 o[NJA][4];
#pragma unroll
        for (int j = 0; j < NJA; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[j][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        float inv[NJA];
#pragma unroll
        for (int j = 0; j < NJA; ++j) inv[j] = 0.f;

        // scores of one K tile in steps of (key fragment, half of d): 3 operand reads (of the NEXT step) dealt between 9 NJ MFMAs (smallest
        // pair products first, fixed order); consecutive MFMAs alternate between the wave's query fragments
        auto scores = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            constexpr int F0 = T * TFK, NFT = (NF - F0 < TFK ? NF - F0 : TFK), NST = 2 * NFT;
            const bf16x8* sk = smem + (size_t)(T % NS) * SLOT * 64 + lane;
            bf16x8 kr[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) kr[0][pl] = sk[pl * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int fl = st >> 1, c = st & 1;
                if (st + 1 < NST && !(ND_AB9_ABL & 16)) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) kr[(st + 1) & 1][pl] = sk[((st + 1) * 3 + pl) * 64];       // block (fl, c) = piece run 3 (2 fl + c)
                }
#define AB9_S(pp, qq)                                                                                                                  \
                _Pragma("unroll") for (int j = 0; j < NJA; ++j)                                                                        \
                    s[j][F0 + fl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[(ND_AB9_ABL & 16) ? 0 : (st & 1)][pp], qr[j][c][qq], s[j][F0 + fl], 0, 0, 0);
                if constexpr (!(ND_AB9_ABL & 4)) { AB9_S(2, 2) AB9_S(2, 1) AB9_S(1, 2) AB9_S(2, 0) AB9_S(0, 2) AB9_S(1, 1) AB9_S(1, 0) AB9_S(0, 1) AB9_S(0, 0) }
#undef AB9_S
                if (st + 1 < NST && !(ND_AB9_ABL & (4 | 16))) {
                    // the next step's three operand reads EARLY in this step's MFMAs (after 1, 2 and 3 x NJ of them): dealt evenly they
                    // put the last read right in front of the next step's lgkmcnt(0) and expose its latency once per step
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NJA, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 6 * NJA, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // exact two-pass softmax over the keys, per query column (lane & 15): in-lane over (f, r), across the 4 lane groups by xor 16 / 32.
        // One fragment at a time (scheduling barriers): left alone, the scheduler interleaves all 26 fragments' exponentials for ILP and
        // spills hundreds of registers at the two-waves-per-SIMD budget.  Only the last fragment can hold keys past N.
        auto softmax = [&]() {
            const float scale = 0.125f;      // 64^-0.5
#pragma unroll
            for (int j = 0; j < NJA; ++j) {
                float mx = -1.0e30f;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = s[j][f][r] * scale;
                        if (f == NF - 1) v = 16 * f + 4 * g + r < N ? v : -1.0e30f;      // finite stand-in for -inf: exp -> exactly 0
                        s[j][f][r] = v;
                        mx = fmaxf(mx, v);
                    }
                    asm volatile("" : "+v"(s[j][f]));           // (ordered like the barriers: keeps instruction selection from batching fragments)
                    __builtin_amdgcn_sched_barrier(0);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pe = nd_exp_neg(s[j][f][r] - mx);
                        s[j][f][r] = pe;
                        sum += pe;
                    }
                    asm volatile("" : "+v"(s[j][f]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                inv[j] = 1.0f / sum;
            }
        };
        // which of a lane's 8 k-slots of the LAST key block hold keys < N (the rest of that V^T block was never written): AND masks per
        // 32-bit register (two bf16 each)
        uint32_t vmask[4];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const int e0 = 2 * w2, e1 = 2 * w2 + 1;
            const int k0 = 32 * (NKB - 1) + 16 * (e0 >> 2) + 4 * g + (e0 & 3), k1 = 32 * (NKB - 1) + 16 * (e1 >> 2) + 4 * g + (e1 & 3);
            vmask[w2] = (k0 < N ? 0x0000ffffu : 0u) | (k1 < N ? 0xffff0000u : 0u);
        }
        // O^T += V^T P over one V^T tile in steps of (key block, d fragment): 3 operand reads (of the next step) between 9 NJ MFMAs; the
        // probabilities of a key block are normalised (as torch: softmax, then @ v) and split into their three bf16 pieces per block
        auto pv = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            constexpr int KB0 = (T - NTK) * TKV, NKT = (NKB - KB0 < TKV ? NKB - KB0 : TKV), NST = 4 * NKT;
            const bf16x8* sv = smem + (size_t)(T % NS) * SLOT * 64 + lane;
            bf16x8 vr[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) vr[0][pl] = sv[pl * 64];                       // (kl = 0, df = 0)
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 pr[NJA][3];
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int kl = st >> 2, df = st & 3, kb = KB0 + kl;
                if (df == 0 && !((ND_AB9_ABL & 2) && st > 0)) {
#pragma unroll
                    for (int j = 0; j < NJA; ++j) {          // the block's probabilities become visible to the compiler only now (see the pin below)
                        asm volatile("" : "+v"(s[j][2 * kb]));
                        if (2 * kb + 1 < NF) asm volatile("" : "+v"(s[j][2 * kb + 1]));
                    }
#pragma unroll
                    for (int j = 0; j < NJA; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int f = 2 * kb + (e >> 2);
                            const float pn = f < NF ? s[j][f < NF ? f : NF - 1][e & 3] * inv[j] : 0.f;
                            // the exact three-piece split of nd_b9_split without its infinity guard: a probability is in [0, 1]
                            const __bf16 h1 = (__bf16)pn;
                            const float r1 = pn - (float)h1;
                            const __bf16 h2 = (__bf16)r1;
                            const __bf16 h3 = (__bf16)(r1 - (float)h2);
                            pr[j][0][e] = h1; pr[j][1][e] = h2; pr[j][2][e] = h3;
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (st + 1 < NST && !(ND_AB9_ABL & 16)) {
                    const int kl1 = (st + 1) >> 2, df1 = (st + 1) & 3;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) vr[(st + 1) & 1][pl] = sv[((df1 * NKT + kl1) * 3 + pl) * 64];
                }
                if (kb == NKB - 1) {           // (no run-time condition here: a branch inside the tile sequence lets the compiler sink the MFMAs below the barriers)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        u32x4 t = __builtin_bit_cast(u32x4, vr[st & 1][pl]);
                        t[0] &= vmask[0]; t[1] &= vmask[1]; t[2] &= vmask[2]; t[3] &= vmask[3];
                        vr[st & 1][pl] = __builtin_bit_cast(bf16x8, t);
                    }
                }
#define AB9_O(pp, qq)                                                                                                                  \
                _Pragma("unroll") for (int j = 0; j < NJA; ++j)                                                                        \
                    o[j][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[(ND_AB9_ABL & 16) ? 0 : (st & 1)][pp], pr[j][qq], o[j][df], 0, 0, 0);
                if constexpr (!(ND_AB9_ABL & 8)) { AB9_O(2, 2) AB9_O(2, 1) AB9_O(1, 2) AB9_O(2, 0) AB9_O(0, 2) AB9_O(1, 1) AB9_O(1, 0) AB9_O(0, 1) AB9_O(0, 0) }
#undef AB9_O
                if (st + 1 < NST && !(ND_AB9_ABL & (8 | 16))) {          // the next step's operand reads early in this step's MFMAs, as in the score steps
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NJA, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 6 * NJA, 0);
                }
                // pin the step's MFMAs HERE: they are pure register operations whose results are only stored at the very end, and instruction
                // selection otherwise emits the whole P V chain behind the last barrier -- with every V^T operand of every tile held (spilled)
                // until then.  An empty volatile asm that takes and returns the accumulators is ordered with the barriers.
#pragma unroll
                for (int j = 0; j < NJA; ++j) asm volatile("" : "+v"(o[j][df]));
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        auto step = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            // everything up to tile T has landed: all that may still be in flight are the pieces of the NS - 1 tiles after it
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(pieces_per_wave(T + 1) + (NS > 2 ? pieces_per_wave(T + 2) : 0) + (NS > 3 ? pieces_per_wave(T + 3) : 0)) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NJ > 0) {
                if constexpr (T < NTK) scores(tc);
                else pv(tc);
            }
            if constexpr (T + NS < NTT) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                          // every wave is done reading slot T % NS
                __builtin_amdgcn_sched_barrier(0);
                stage(std::integral_constant<int, T + NS>{});
            }
            if constexpr (NJ > 0 && T == NTK - 1 && !(ND_AB9_ABL & 1)) softmax();          // under the landing of the first V^T tiles
        };
        run_steps(step, std::make_integer_sequence<int, NTT>{});

        if constexpr (NJ > 0) {
            // o[j][df][r] = O[q = 16 qf + (lane & 15)][d = 16 df + 4 g + r]
            const int Cm = heads * 64;
#pragma unroll
            for (int j = 0; j < NJA; ++j) {
                const int qo = (qf0 + wave + 4 * j) * 16 + li;
                if (qo < N) {
#pragma unroll
                    for (int df = 0; df < 4; ++df) {
                        if (split_out)
                            nd_b9_store4(reinterpret_cast<bf16x8*>(out), Cm >> 5, b * N + qo, hd * 64 + 16 * df + 4 * g, o[j][df][0], o[j][df][1], o[j][df][2],
                                         o[j][df][3]);
                        else
                            *reinterpret_cast<float4*>(out + ((size_t)b * N + qo) * Cm + (size_t)hd * 64 + 16 * df + 4 * g) =
                                make_float4(o[j][df][0], o[j][df][1], o[j][df][2], o[j][df][3]);
                    }
                }
            }
        }
    };
    if (njw == 2) body(std::integral_constant<int, 2>{});
    else if (njw == 1) body(std::integral_constant<int, 1>{});
    else body(std::integral_constant<int, 0>{});
}

template <int NF>
static hipError_t launch_attention_b9(const void* att, float* out, int B, int N, int heads, int split_out, hipStream_t st) {
    const int QG = (NF + 7) / 8;
    constexpr int NKB = (NF + 1) / 2, TFK = NF < ND_AB9_TFK ? NF : ND_AB9_TFK, TKV = NKB < ND_AB9_TKV ? NKB : ND_AB9_TKV;
    constexpr size_t lds = (size_t)ND_AB9_NS * (TFK * 6 > TKV * 12 ? TFK * 6 : TKV * 12) * 1024;
    if (lds > 64 * 1024) {
        hipError_t e = nd_allow_dynamic_lds((const void*)k_attention_b9<NF>, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_attention_b9<NF>), dim3(B * heads * QG), dim3(256), lds, st, (const bf16x8*)att, out, B, N, heads, QG, split_out);
    return hipGetLastError();
}

hipError_t nd_launch_attention_b9(const void* att, float* out, int B, int N, int heads, int split_out, hipStream_t st) {
    switch ((N + 15) / 16) {
#define AB_CASE(NFV) case NFV: return launch_attention_b9<NFV>(att, out, B, N, heads, split_out, st);
        AB_CASE(1) AB_CASE(2) AB_CASE(3) AB_CASE(4) AB_CASE(5) AB_CASE(6) AB_CASE(7) AB_CASE(8)
        AB_CASE(9) AB_CASE(10) AB_CASE(11) AB_CASE(12) AB_CASE(13) AB_CASE(14) AB_CASE(15) AB_CASE(16)
#undef AB_CASE
    }
    return hipErrorInvalidValue;
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
