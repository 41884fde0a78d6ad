// nd_common.hpp -- shared device code for libnd_hip.so (gfx950 / CDNA4 only).
//
// Skinny-M weight-streaming GEMM building blocks.  Every Linear on the sampling path has a small
// row count M (B*mc images, 32 at the headline config) against 4096..150528-wide weights, so the
// kernels are bound by streaming W once from HBM; the f32-input MFMA (v_mfma_f32_16x16x4_f32, exact
// f32) keeps the arithmetic off the VALU and just keeps up with the stream at M = 32.
//
// DATA LAYOUT ("frag16" packing).  Both GEMM operands are K-contiguous matrices A[R][K].  They are
// stored as 16-row x 16-column blocks of 1 KiB, block (r/16, k/16) at float offset
// ((r/16)*(K/16) + k/16)*256, and inside a block element (r, k) at ((r%16) + 16*((k%16)/4))*4 + k%4:
// exactly the order in which the 64 lanes of a wave consume it (lane l = row l&15, k-quad l>>4), so
// every wave-level load is one fully coalesced 1 KiB read and a workgroup streams one contiguous
// region.  Weights are packed once at load time; activations are written packed by the producing
// kernel's epilogue.  Measured on MI355X (tools/ubench_skinny.hip, 5 members x 67 MB): row-major
// operands 2.5 TB/s, packed 3.9 TB/s, packed + nontemporal W loads 4.2 TB/s (plain read: 5.4-6.0).
#pragma once
#include <type_traits>
#include <mutex>
#include <unordered_map>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define ND_ACT_NONE 0
#define ND_ACT_SOFTPLUS 1
#define ND_ACT_RELU 2
#define ND_ACT_GELU 3

__device__ __forceinline__ float nd_softplus(float x) {
    // torch softplus(beta=1, threshold=20): x if x > 20 else log1p(exp(x)).
    // Evaluated as max(x,0) + log1p(u), u = exp(-|x|) in (0,1], with the compensated form
    // log1p(u) = log(w) - ((w-1)-u)/w, w = fl(1+u)  (1-2 ulp; the correction term is ~1e-8, so an approximate reciprocal
    // is enough).  Half the instructions of log1pf(expf(x)) (whose log1pf goes through f64 here) -- this function is the
    // bulk of the step kernels' epilogue, which runs on one wave per SIMD with nothing to overlap it.
    // The operation sequence is pinned (no contraction left to the compiler, the one FMA explicit): every kernel that evaluates it
    // returns the same bits for the same argument.
#pragma clang fp contract(off)
    const float u = expf(-fabsf(x));
    const float w = 1.0f + u;
    const float c = (w - 1.0f) - u;
    const float r = fmaxf(x, 0.0f) + __builtin_fmaf(-c, __builtin_amdgcn_rcpf(w), logf(w));
    return x > 20.0f ? x : r;
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

// erf(a) to < 1 ulp of the exact value (measured against fp64 over [-6, 6] with a correctly rounded exp: 0.97 ulp; the device exp above adds
// at most 1 ulp of exp(r) <= 0.4 to the tail branch), branch-free: both of N. Juffa's minimax forms (|a| <= 0.9277: a + a P(a^2); beyond:
// 1 - exp(Q(|a|))) are evaluated and one is selected -- 15 FMAs and one v_exp_f32 per value, where the library erff takes a data-dependent
// branch per lane group.  Used by the exact-erf GELU of timm's Mlp (the fc1 epilogue: 19 M values per ViT block at B = 32).
__device__ __forceinline__ float nd_erf(float a) {
    const float t = fabsf(a), s = a * a;
    float r = __builtin_fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = __builtin_fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = __builtin_fmaf(r, s, u);
    r = __builtin_fmaf(r, t, -1.06777877e-1f);
    r = __builtin_fmaf(r, t, -6.34846687e-1f);
    r = __builtin_fmaf(r, t, -1.28717512e-1f);
    r = __builtin_fmaf(r, t, -t);
    const float big = __builtin_copysignf(1.0f - nd_exp_neg(fmaxf(r, -100.0f)), a);      // (the clamp keeps the argument finite for huge |a|)
    float q = -5.96761703e-4f;
    q = __builtin_fmaf(q, s, 4.99119423e-3f);
    q = __builtin_fmaf(q, s, -2.67681349e-2f);
    q = __builtin_fmaf(q, s, 1.12819925e-1f);
    q = __builtin_fmaf(q, s, -3.76125336e-1f);
    q = __builtin_fmaf(q, s, 1.28379166e-1f);
    const float small = __builtin_fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}

__device__ __forceinline__ float nd_act(float v, int act) {
    switch (act) {
        case ND_ACT_SOFTPLUS: return nd_softplus(v);
        case ND_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case ND_ACT_GELU: return 0.5f * v * (1.0f + nd_erf(v * 0.70710678118654752440f));
        default: return v;
    }
}

// float offset of element (r, k) of a frag16-packed matrix with nch = K/16 chunks per row
__host__ __device__ __forceinline__ size_t nd_pk(int r, int k, int nch) {
    return ((size_t)(r >> 4) * nch + (k >> 4)) * 256 + (size_t)(((r & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3));
}
static inline size_t nd_packed_floats(int R, int K) { return (size_t)((R + 15) / 16) * 16 * (size_t)K; }

// HALF-PRECISION OPERANDS ("frag32h", the fp16 mode).  Same idea with 16-row x 32-column blocks of fp16, again 1 KiB:
// block (r/16, k/32) at BYTE offset ((r/16)*(K/32) + k/32)*1024, element (r, k) inside it at half index
// ((r%16) + 16*((k%32)/8))*8 + k%8 -- lane l of a wave holds the 8 halfs k = 8*(l>>4)..+7 of row l&15, the operand
// shape of v_mfma_f32_16x16x32_f16.  A block is 256 float-sized words like a frag16 block, so the streaming kernels
// address both forms identically with nch = K/32 instead of K/16.  Products are exact, accumulation is fp32.
__host__ __device__ __forceinline__ size_t nd_pkh(int r, int k, int nch32) {   // half index
    return ((size_t)(r >> 4) * nch32 + (k >> 5)) * 512 + (size_t)(((r & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7));
}
static inline size_t nd_packed_bytes_dt(int R, int K, int half) {
    return (size_t)((R + 15) / 16) * 16 * (size_t)K * (half ? 2 : 4);
}

// row-major fp32 [R][K] -> frag32h (round to nearest even); rows R .. 16*ceil(R/16) zero-filled.  16 B per thread.
static __global__ __launch_bounds__(256) void k_pack_rows_h(const float* __restrict__ src, _Float16* __restrict__ dst, int R, int K) {
    const int nch = K >> 5;
    const size_t total = (size_t)((R + 15) / 16) * nch * 64;   // 16-byte pieces
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int rt = (int)(blk / nch), c = (int)(blk % nch);
        const int r = rt * 16 + (lane & 15);
        f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < R) {
            const float* p = src + (size_t)r * K + c * 32 + 8 * (lane >> 4);
            const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            h = f16x8{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
        }
        reinterpret_cast<f16x8*>(dst)[i] = h;
    }
}

// frag32h [R][K] -> row-major fp32 (tests / debugging)
static __global__ __launch_bounds__(256) void k_unpack_rows_h(const _Float16* __restrict__ src, float* __restrict__ dst, int R, int K) {
    const int nch = K >> 5;
    const size_t total = (size_t)((R + 15) / 16) * nch * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int rt = (int)(blk / nch), c = (int)(blk % nch);
        const int r = rt * 16 + (lane & 15);
        if (r < R) {
            const f16x8 h = reinterpret_cast<const f16x8*>(src)[i];
            float* p = dst + (size_t)r * K + c * 32 + 8 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = (float)h[j];
        }
    }
}

// row-major [R][K] -> frag16; rows R .. 16*ceil(R/16) are zero-filled.  One float4 per thread.
static __global__ __launch_bounds__(256) void k_pack_rows(const float* __restrict__ src, float* __restrict__ dst, int R, int K) {
    const int nch = K >> 4;
    const size_t total = (size_t)((R + 15) / 16) * nch * 64;   // float4 count
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int rt = (int)(blk / nch), c = (int)(blk % nch);
        const int r = rt * 16 + (lane & 15);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) v = *reinterpret_cast<const float4*>(src + (size_t)r * K + c * 16 + 4 * (lane >> 4));
        reinterpret_cast<float4*>(dst)[i] = v;
    }
}

// frag16 [R][K] -> row-major (tests / debugging)
static __global__ __launch_bounds__(256) void k_unpack_rows(const float* __restrict__ src, float* __restrict__ dst, int R, int K) {
    const int nch = K >> 4;
    const size_t total = (size_t)((R + 15) / 16) * nch * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int rt = (int)(blk / nch), c = (int)(blk % nch);
        const int r = rt * 16 + (lane & 15);
        if (r < R) *reinterpret_cast<float4*>(dst + (size_t)r * K + c * 16 + 4 * (lane >> 4)) = reinterpret_cast<const float4*>(src)[i];
    }
}

__device__ __forceinline__ f16x8 nd_as_h8(const float4& v) {   // the 16 bytes of a lane's operand, viewed as 8 halfs
    return __builtin_bit_cast(f16x8, v);
}

// 16-byte load from GLOBAL memory (address space 1: a pointer read out of a descriptor table is generic to the compiler
// and would become flat_load, which also counts on lgkmcnt).  NT = nontemporal (streamed once).
typedef __attribute__((address_space(1))) const f32x4 nd_gf4;
template <bool NT>
__device__ __forceinline__ float4 nd_ld16(const float* p) {
    const nd_gf4* g = (const nd_gf4*)p;
    const f32x4 v = NT ? __builtin_nontemporal_load(g) : *g;
    return make_float4(v[0], v[1], v[2], v[3]);
}

#ifdef ND_WG_TIMING
__device__ long long* nd_dbg_times = nullptr;   // tools/ubench_skinny.hip: per-workgroup (start, loop end, end) clocks
#endif

// scalar load from GLOBAL memory through a pointer the compiler only knows as generic (it came out of a descriptor table):
// a pending flat_load cannot be counted (it may be LDS or global), so every later wait would become vmcnt(0) lgkmcnt(0)
__device__ __forceinline__ float nd_ldg(const float* p) {
    return *(const __attribute__((address_space(1))) float*)p;
}

// One skinny Linear on packed operands, for `nm` members that share the layer shape (K, N):
//   MODE 0: out[m,n] = act(scale[t,n] * sum_k x[m,k] w[n,k] + shift[t,n])
//   MODE 1: that value is not stored; its projection onto C rows is: part[m,c,tile] = sum_{n in tile} pw[c,n]*v
//           -- lin3 + unetnorm3 + softplus + lin4 (latent_model.py:181-184) in one pass.
//   MODE 2: split-K partial sums, no epilogue: part[slab, m, n] = sum_{k in slab} x[m,k] w[n,k]
#ifndef ND_PF_TWO
#define ND_PF_TWO 0
#endif
#ifndef ND_PF_LEAD
#define ND_PF_LEAD 0            // register stages between the cross-launch prefetch and the end of a wave's stream (~0.45 us each at M = 32)
#endif
struct SkinnyDesc {
    const float* x;      // frag16 [M][K]   (frag32h in the fp16 kernels: opaque 1 KiB blocks either way)
    const float* w;      // frag16 [N][K]   (nn.Linear weight, rows padded to 16 with zeros)
    const float* scale;  // [rows, N] or nullptr (=1)
    const float* shift;  // [rows, N] or nullptr (=0)
    float* out;          // MODE 0: out_packed 0 = row-major fp32 [M][N], 1 = frag16 fp32, 2 = frag32h fp16 (N % 32 == 0)
    const float* pw;     // [C, N]            (MODE 1)
    float* part;         // MODE 1: [M, C, ceil(N/16)];  MODE 2: [S, Mpad, Npad]
    int K, N, C, act, out_packed;
    int keep;            // 1: this member's W is read with default-policy loads in an NT kernel (kept in the Infinity Cache across steps)
    const float* pf;     // or null: the weight matrix (same shape, same packing) of the NEXT launch of this member's step chain: near the end
                         // of its stream every wave touches the lines of the first register stage the same workgroup of that launch will ask
                         // for -- same blockIdx, hence same XCD and L2 -- so that launch's ramp starts on L2 hits instead of HBM latency
};

// Up to ND_INLINE_DESCS members' descriptors travel BY VALUE in the kernel arguments (table == nullptr): a workgroup then has its
// operand pointers after one scalar load from the kernarg segment instead of a kernarg load followed by a dependent global load
// of table[g] -- one memory round trip less before the first weight request of every launch.
#define ND_INLINE_DESCS 8
struct SkinnyInline { SkinnyDesc d[ND_INLINE_DESCS]; };

// struct copy out of the constant address space (scalar loads when the address is wave-uniform), word by word: the implicit copy
// constructor cannot bind an address-space-qualified source
template <typename T>
__device__ __forceinline__ T nd_ldc(const __attribute__((address_space(4))) void* p) {
    static_assert(sizeof(T) % 4 == 0, "word-sized structs only");
    T out;
    const __attribute__((address_space(4))) uint32_t* w = (const __attribute__((address_space(4))) uint32_t*)p;
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) o[i] = w[i];
    return out;
}

// WORK DECOMPOSITION.  A workgroup never mixes members (it would have to stream two activation matrices and finishes
// 40 % late: measured).  Each member's nfr = ceil(N/16) 16-column output fragments go to wpm = gridDim.x / nm workgroups:
// base = nfr / wpm each, the first nfr % wpm of them one more.  NF (template) = the larger count; a workgroup holding
// NF - 1 fragments skips the last register slot (uniform branch).  The launcher picks wpm so that the grid is at most one
// workgroup per CU.  grid.y = 16*MT-row groups; grid.z = k-slabs (MODE 2 only).
// A wave keeps NF weight fragments and MT activation fragments per k-chunk in registers, so an activation fragment is
// loaded once per NF weight fragments.  The WAVES waves of a workgroup split K (interleaved groups of U chunks) and are
// summed through LDS in a fixed order => bitwise reproducible.
// MFMA 16x16x4 f32: A[i=l&15][k=l>>4] <- W rows, B[k=l>>4][j=l&15] <- x rows, D[i=4*(l>>4)+r][j=l&15];
// lane l's float4 holds k = 4*(l>>4)..+3 of a chunk and element jj feeds MFMA jj (same k permutation on
// both operands).
// H = 1: operands are frag32h (fp16), one v_mfma_f32_16x16x32_f16 per fragment pair and 32-column chunk.
#define ND_SKINNY_STATIC_LDS (60 * 1024)
__host__ __device__ constexpr int nd_skinny_red_bytes(int waves, int nf, int mt) { return waves * nf * mt * 1024; }
template <int MT, int NF, int WAVES, int U, int MODE, bool NT, int H = 0>
__global__ __launch_bounds__(WAVES * 64) void k_skinny(SkinnyInline di, const SkinnyDesc* __restrict__ table, int nm,
                                                       int M, int t, int cps) {
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef ND_WG_TIMING
    const long long dbg_t0 = wall_clock64();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps stream addresses in SGPRs
    // both sources are read through the CONSTANT address space (scalar loads; a select between a kernarg and a generic pointer
    // would turn every descriptor access into a flat vector load): `di` is the first kernel argument = kernarg offset 0
    typedef const __attribute__((address_space(4))) char* nd_cbytes;
    const nd_cbytes dsrc = table ? (nd_cbytes)(uintptr_t)table : (nd_cbytes)__builtin_amdgcn_kernarg_segment_ptr();
    const int K = nd_ldc<SkinnyDesc>(dsrc).K, N = nd_ldc<SkinnyDesc>(dsrc).N;
    const int nch = H ? K >> 5 : K >> 4, nfr = (N + 15) >> 4, mtiles = (M + 15) >> 4;
    const int wpm = gridDim.x / nm;                               // workgroups per member
    const int g = blockIdx.x / wpm, j = blockIdx.x - g * wpm;     // member, workgroup inside the member
    const int base = nfr / wpm, rem = nfr - base * wpm;
    const int nact = base + (j < rem ? 1 : 0);                    // fragments this workgroup owns: NF or NF - 1
    const int fi0 = j * base + min(j, rem);                       // its first fragment inside the member
    const SkinnyDesc d = nd_ldc<SkinnyDesc>(dsrc + (size_t)g * sizeof(SkinnyDesc));
    const int c0 = MODE == 2 ? blockIdx.z * cps : 0;
    const int c1 = MODE == 2 ? min(c0 + cps, nch) : nch;
    const float* wp[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) wp[f] = d.w + ((size_t)min(fi0 + f, nfr - 1) * nch + c0) * 256;   // uniform; lane term added at the load
    const float* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = d.x + ((size_t)min((int)blockIdx.y * MT + mt, mtiles - 1) * nch + c0) * 256;
    f32x4 acc[NF][MT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Epilogue operands are requested NOW (registers; parked in LDS after the main loop): the scale/shift entries of the
    // workgroup's output columns and the lin4 rows of MODE 1.
    __shared__ float ssc[NF][16], ssh[NF][16];
    __shared__ float pws[MODE == 1 ? NF : 1][MODE == 1 ? 8 : 1][16];
    const int eact = d.act, eC = d.C, epacked = d.out_packed;
    float r_sc = 1.0f, r_sh = 0.0f, r_pw = 0.0f;
    if (MODE != 2) {
        // thread tid < NF*16 owns column (f = tid/16, nl = tid%16); thread tid < NF*C*16 owns lin4 entry (f, c, nl)
        if (tid < NF * 16) {
            const int n = min(min(fi0 + (tid >> 4), nfr - 1) * 16 + (tid & 15), N - 1);
            if (d.scale) r_sc = nd_ldg(d.scale + (size_t)t * N + n);
            if (d.shift) r_sh = nd_ldg(d.shift + (size_t)t * N + n);
        }
        // lin4 rows: one entry per thread when they fit (NF*C*16 <= threads), requested now and parked after the main loop;
        // otherwise (large C) fetched after the loop.  No load inside a loop of unknown trip count here: with one pending the
        // compiler can no longer count outstanding loads and degrades every wait of the main loop to vmcnt(0).
        // (an unconditional load on a clamped address, in the global address space: see nd_ldg)
        if (MODE == 1) {
            const int e = min(tid, NF * eC * 16 - 1);
            const int f = e / (eC * 16), c = (e / 16) % eC, nl = e & 15;
            const int n = min(min(fi0 + f, nfr - 1) * 16 + nl, N - 1);
            r_pw = nd_ldg(d.pw + (size_t)c * N + n);
        }
    }

    // cross-launch prefetch (d.pf): lane L of instruction q touches line (q*64 + L) % 8 of chunk (q*64 + L) / 8 of this wave's first
    // group in the next launch -- chunk ci = (fragment ci / U, chunk wave*U + ci % U of that fragment's row), 8 lines of 128 B each
    const float* pfp[2] = {nullptr, nullptr};
    if (MODE != 2 && NT && d.pf) {      // NT: the launch streams more than the Infinity Cache keeps (small launches run out of it: K = 1 measured 1.3 % slower with the prefetch)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = q * 64 + lane, ci = min(idx >> 3, nact * U - 1), f = ci / U, u = ci - f * U;
            pfp[q] = d.pf + ((size_t)min(fi0 + f, nfr - 1) * nch + min(wave * U + u, nch - 1)) * 256 + (idx & 7) * 32;
        }
    }
    float pfv = 0.f;
    const int nck = max(c1 - c0, 0);
    const int ngroups = nck / U;     // full groups of U chunks
    const int ngw = ngroups > wave ? (ngroups - wave + WAVES - 1) / WAVES : 0;
    const int glast = ngroups > 0 ? ngroups - 1 : 0;

    // MAIN LOOP.  Two register stages (A, B), each a whole group of U chunks (NF weight + MT activation fragments), used
    // alternately with no copies: while the MFMAs of the current group run, the loads of the next one are issued, so every
    // wave keeps one group in flight under its arithmetic.  Scheduling barriers pin that order (left alone, the compiler
    // sinks the loads below the MFMAs to save registers and then waits on them at once).
    // FULL = false (a workgroup with NF - 1 fragments): the last fragment slot is neither loaded nor multiplied.
    const int lane4 = lane * 4;
    auto run = [&](auto ntc, auto fullc) {
        constexpr bool NTV = decltype(ntc)::value, FULL = decltype(fullc)::value;
        constexpr int NFA = FULL ? NF : (NF > 1 ? NF - 1 : 1);         // active fragment slots
        float4 wA[U][NFA], xA[U][MT], wB[U][NFA], xB[U][MT];
        auto LD = [&](float4 (&w)[U][NFA], float4 (&x)[U][MT], int grp) {
            const size_t go = (size_t)grp * U * 256;
#ifdef ND_SKINNY_ABLATE_W      // timing ablation only (variant library; results WRONG): every weight load of a wave goes to its FIRST register stage
            const size_t gow = (size_t)min(wave, glast) * U * 256;     // -- cache hits instead of fabric traffic, same instructions, same MFMAs
#else
            const size_t gow = go;
#endif
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int f = 0; f < NFA; ++f) w[u][f] = nd_ld16<NTV>(wp[f] + gow + u * 256 + lane4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) x[u][mt] = nd_ld16<false>(xp[mt] + go + u * 256 + lane4);
            }
        };
        auto MMA = [&](const float4 (&w)[U][NFA], const float4 (&x)[U][MT]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int jj = 0; jj < (H ? 1 : 4); ++jj)
#pragma unroll
                    for (int f = 0; f < NFA; ++f)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (H) {
                                acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(nd_as_h8(w[u][f]), nd_as_h8(x[u][mt]), acc[f][mt], 0, 0, 0);
                            } else {
                                const float wv = jj == 0 ? w[u][f].x : jj == 1 ? w[u][f].y : jj == 2 ? w[u][f].z : w[u][f].w;
                                const float xv = jj == 0 ? x[u][mt].x : jj == 1 ? x[u][mt].y : jj == 2 ? x[u][mt].z : x[u][mt].w;
                                acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[f][mt], 0, 0, 0);
                            }
                        }
        };
        auto G = [&](int i) { return min(wave + i * WAVES, glast); };      // group of this wave's i-th turn (clamped)
        int i = 0;
        if (ngw > 0) LD(wA, xA, G(0));
        // Steady state.  The next stage's loads are spread through the current stage's MFMAs (one load per MR MFMAs), so the
        // matrix pipe never waits for the ~150 cycles of load issue: -3 us per launch against issuing them in a block first.
        constexpr int NL = U * (NFA + MT), NM = U * (H ? 1 : 4) * NFA * MT, MR = NM / NL > 0 ? NM / NL : 1;
#define ND_MIX()                                                                                     \
        _Pragma("unroll") for (int q_ = 0; q_ < NL; ++q_) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, MR, 0);                                      \
        }                                                                                            \
        __builtin_amdgcn_sched_barrier(0);
        const int pf_at = ngw > ND_PF_LEAD ? (ngw - ND_PF_LEAD) & ~1 : 0;        // ~ND_PF_LEAD register stages before the end of the stream
        for (; i + 1 < ngw; i += 2) {
            if (MODE != 2 && pfp[0] && i == pf_at && ND_PF_LEAD > 0) {
                pfv += nd_ldg(pfp[0]) + nd_ldg(pfp[1]);
                if (ND_PF_TWO) pfv += nd_ldg(pfp[0] + (size_t)WAVES * U * 256) + nd_ldg(pfp[1] + (size_t)WAVES * U * 256);     // the wave's second group too
            }
            LD(wB, xB, G(i + 1));
            MMA(wA, xA);
            ND_MIX()
            LD(wA, xA, G(i + 2));          // i + 2 == ngw on the last pair of an even count: re-reads a valid group, unused
            MMA(wB, xB);
            ND_MIX()
        }
#undef ND_MIX
        if (ND_PF_LEAD == 0 && MODE != 2 && pfp[0]) pfv += nd_ldg(pfp[0]) + nd_ldg(pfp[1]);      // (variant: behind the last stage's loads)
        if (i < ngw) MMA(wA, xA);
    };
    const bool nt_here = NT && !d.keep;
    if (nact == NF) {
        if (nt_here) run(std::true_type{}, std::true_type{});
        else run(std::false_type{}, std::true_type{});
    } else {
        if (nt_here) run(std::true_type{}, std::false_type{});
        else run(std::false_type{}, std::false_type{});
    }
    // leftover chunks (chunk count not a multiple of U): chunk c goes to wave c % WAVES
    for (int c = ngroups * U + wave; c < nck; c += WAVES) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float4 w4 = nd_ld16<false>(wp[f] + (size_t)c * 256 + lane4);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 x4 = nd_ld16<false>(xp[mt] + (size_t)c * 256 + lane4);
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
                if (H) acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(nd_as_h8(w4), nd_as_h8(x4), acc[f][mt], 0, 0, 0);
                else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[jj], xv[jj], acc[f][mt], 0, 0, 0);
                }
            }
        }
    }
#ifdef ND_WG_TIMING
    const long long dbg_t1 = wall_clock64();
#endif
    if (ND_PF_LEAD < 0 && MODE != 2 && pfp[0]) {                                  // (variant: at the start of the epilogue)
        pfv += nd_ldg(pfp[0]) + nd_ldg(pfp[1]);
        if (ND_PF_TWO) pfv += nd_ldg(pfp[0] + (size_t)WAVES * U * 256) + nd_ldg(pfp[1] + (size_t)WAVES * U * 256);
    }
    // park the epilogue operands (in registers since the prologue) in LDS; the first barrier below publishes them
    if (MODE != 2) {
        if (tid < NF * 16) { ssc[tid >> 4][tid & 15] = r_sc; ssh[tid >> 4][tid & 15] = r_sh; }
        if (MODE == 1) {
            if (NF * eC * 16 <= WAVES * 64) {
                if (tid < NF * eC * 16) {
                    const int f = tid / (eC * 16), nl = tid & 15;
                    pws[f][(tid / 16) % eC][nl] = (min(fi0 + f, nfr - 1) * 16 + nl < N) ? r_pw : 0.f;
                }
            } else {
                for (int e = tid; e < NF * eC * 16; e += WAVES * 64) {
                    const int f = e / (eC * 16), c = (e / 16) % eC, nl = e & 15;
                    const int n = min(fi0 + f, nfr - 1) * 16 + nl;
                    pws[f][c][nl] = n < N ? nd_ldg(d.pw + (size_t)c * N + n) : 0.f;
                }
            }
        }
    }

    // ---- all fragments at once: cross-wave reduction (fixed order) + epilogue, two barriers in total ----
    // red[w] holds wave w's accumulators in MFMA D order, element ((f*MT + mt)*4 + r)*64 + l  <->  n = 4*(l>>4)+r, m = l&15.
    // The reduced, activated value replaces plane 0 in place (same thread reads the four planes and writes plane 0).
    // (above 60 KiB -- five or six weight fragments against four or five row fragments -- the buffer is dynamic LDS: the launcher
    //  passes nd_skinny_dynlds<>() bytes; the smaller shapes keep their static array and their code)
    constexpr bool DYN = nd_skinny_red_bytes(WAVES, NF, MT) > ND_SKINNY_STATIC_LDS;
    __shared__ __attribute__((aligned(16))) float red_static[DYN ? 1 : WAVES][DYN ? 1 : NF * MT * 256];
    extern __shared__ __attribute__((aligned(16))) float red_dynamic[];
    float (*const red)[NF * MT * 256] = DYN ? reinterpret_cast<float (*)[NF * MT * 256]>(red_dynamic)
                                            : reinterpret_cast<float (*)[NF * MT * 256]>(&red_static[0][0]);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][((f * MT + mt) * 4 + r) * 64 + lane] = acc[f][mt][r];
    __syncthreads();
#ifdef ND_WG_TIMING
    const long long dbg_t2 = wall_clock64();
    long long dbg_t3 = dbg_t2;
#endif
    float* const R = red[0];
    const int m0 = blockIdx.y * 16 * MT;
    if (MODE == 2) {
        // raw partial sums, [slab][Mp][Np]; lane l of a reduced tile owns n = 4*(l>>4)..+3 of row m = l&15
        const int Mp = mtiles * 16, Np = nfr * 16;
        for (int e = tid; e < nact * MT * 64; e += WAVES * 64) {
            const int f = e / (MT * 64), mt = (e >> 6) % MT, l = e & 63;
            const int mtg = blockIdx.y * MT + mt;
            if (mtg < mtiles) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = ((f * MT + mt) * 4 + r) * 64 + l;
                    float sum = red[0][q];
#pragma unroll
                    for (int w = 1; w < WAVES; ++w) sum += red[w][q];
                    v[r] = sum;
                }
                *reinterpret_cast<float4*>(d.part + ((size_t)blockIdx.z * Mp + mtg * 16 + (l & 15)) * Np + (fi0 + f) * 16 + 4 * (l >> 4)) =
                    make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        // fully unrolled, no branches: with one wave per SIMD the only latency hiding is the instruction-level parallelism
        // of these NF*MT*256/threads independent activations (slots f >= nact hold zeros: computed, never stored further)
        static_assert((MT * 256) % (WAVES * 64) == 0, "epilogue assumes MT*256 is a multiple of the workgroup size");
        constexpr int EPF = MT * 256 / (WAVES * 64);        // elements per thread and fragment
        float ev[NF][EPF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int jq = 0; jq < EPF; ++jq) {
                const int q = f * MT * 256 + jq * WAVES * 64 + tid;
                float sum = red[0][q];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) sum += red[w][q];
                const int nl = 4 * ((q & 63) >> 4) + ((q >> 6) & 3);
                ev[f][jq] = ssc[f][nl] * sum + ssh[f][nl];
            }
        // the activation is uniform: branch once, then a straight run of independent evaluations
        auto activate = [&](auto fn) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int jq = 0; jq < EPF; ++jq) ev[f][jq] = fn(ev[f][jq]);
        };
        switch (eact) {
            case ND_ACT_SOFTPLUS: activate([](float v) { return nd_softplus(v); }); break;
            case ND_ACT_RELU: activate([](float v) { return v > 0.0f ? v : 0.0f; }); break;
            case ND_ACT_GELU: activate([](float v) { return 0.5f * v * (1.0f + nd_erf(v * 0.70710678118654752440f)); }); break;
            default: break;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int jq = 0; jq < EPF; ++jq) {
                const int q = f * MT * 256 + jq * WAVES * 64 + tid;
                const int nl = 4 * ((q & 63) >> 4) + ((q >> 6) & 3);
                R[q] = ((fi0 + f) * 16 + nl < N) ? ev[f][jq] : 0.f;
            }
        __syncthreads();
#ifdef ND_WG_TIMING
        dbg_t3 = wall_clock64();
#endif
        if (MODE == 0) {
            if (epacked == 2) {
                // frag32h: a fragment's 16 columns are k-groups (nfi&1)*2 + {0,1} of block (m-tile, nfi/2): lanes
                // 32*(nfi&1) .. +31 of it, 512 contiguous bytes per m-tile
                _Float16* outh = reinterpret_cast<_Float16*>(d.out);
                const int nch_o = N >> 5;
                for (int e = tid; e < nact * MT * 32; e += WAVES * 64) {
                    const int f = e / (MT * 32), mt = (e >> 5) % MT, L = e & 31, nfi = fi0 + f;
                    const int mtg = blockIdx.y * MT + mt;
                    if (mtg < mtiles) {
                        const float* tp = R + (f * MT + mt) * 256 + (L & 15) + 32 * (L >> 4);   // l0 = m + 16*(2*(L>>4)), l1 = l0 + 16
                        const f16x8 h = {(_Float16)tp[0], (_Float16)tp[64], (_Float16)tp[128], (_Float16)tp[192],
                                         (_Float16)tp[16], (_Float16)tp[80], (_Float16)tp[144], (_Float16)tp[208]};
                        *reinterpret_cast<f16x8*>(outh + ((size_t)mtg * nch_o + (nfi >> 1)) * 512 + ((nfi & 1) * 32 + L) * 8) = h;
                    }
                }
            } else if (epacked) {
                // the 16x16 block (m-tile, fragment) is one contiguous 1 KiB of the frag16 output
                for (int e = tid; e < nact * MT * 64; e += WAVES * 64) {
                    const int f = e / (MT * 64), mt = (e >> 6) % MT, L = e & 63;
                    const int mtg = blockIdx.y * MT + mt;
                    if (mtg < mtiles) {
                        const float* tp = R + (f * MT + mt) * 256 + L;
                        *reinterpret_cast<float4*>(d.out + ((size_t)mtg * nfr + fi0 + f) * 256 + L * 4) = make_float4(tp[0], tp[64], tp[128], tp[192]);
                    }
                }
            } else {
                for (int e = tid; e < nact * MT * 256; e += WAVES * 64) {
                    const int f = e / (MT * 256), ml = (e >> 4) % (16 * MT), nl = e & 15;
                    const int m = m0 + ml, n = (fi0 + f) * 16 + nl;
                    if (m < M && n < N) d.out[(size_t)m * N + n] = R[((f * MT + (ml >> 4)) * 4 + (nl & 3)) * 64 + (ml & 15) + 16 * (nl >> 2)];
                }
            }
        } else {
            for (int e = tid; e < nact * 16 * MT * eC; e += WAVES * 64) {
                const int f = e / (16 * MT * eC), ml = (e / eC) % (16 * MT), c = e % eC;
                const int m = m0 + ml;
                if (m < M) {
                    const float* tp = R + (f * MT + (ml >> 4)) * 256 + (ml & 15);
                    float sum = 0.f;
#pragma unroll
                    for (int nl = 0; nl < 16; ++nl) sum += pws[MODE == 1 ? f : 0][MODE == 1 ? c : 0][nl] * tp[(nl & 3) * 64 + 16 * (nl >> 2)];
                    d.part[((size_t)m * eC + c) * nfr + fi0 + f] = sum;
                }
            }
        }
    }
    // the prefetched values themselves are of no interest: a never-true use keeps the loads (and the compiler's own count of them) alive
    if (MODE != 2 && pfv == 1.2345678e30f && d.part) d.part[0] = pfv;
#ifdef ND_WG_TIMING
    if (tid == 0 && nd_dbg_times) {
        long long* q = nd_dbg_times + ((size_t)MODE * 8192 + blockIdx.x) * 3;      // one region per MODE: the last launch of each stays
        q[0] = dbg_t0; q[1] = dbg_t1; q[2] = wall_clock64();
        long long* q2 = nd_dbg_times + ((size_t)MODE * 8192 + 4096 + blockIdx.x) * 3;
        q2[0] = dbg_t2; q2[1] = dbg_t3; q2[2] = 0;
    }
#endif
}

// out[m,n] = act(scale[n] * sum_s part[s,m,n] + shift[n]); slabs summed in order (reproducible).
// One thread per 4 consecutive n; output frag16 (feeds the next skinny GEMM) or row-major.
struct SplitKEpiDesc {
    const float* part;   // [S, Mpad, Npad]
    const float* scale;  // [N] or nullptr
    const float* shift;  // [N] or nullptr
    float* out;          // frag16 [M][N] (N % 16 == 0) or row-major [M][N]
    int N, S, act, out_packed;
};

static __global__ __launch_bounds__(256) void k_splitk_epilogue(SplitKEpiDesc d0, const SplitKEpiDesc* __restrict__ table, int M, int S) {
    const SplitKEpiDesc d = table ? table[blockIdx.z] : d0;
    const int N = d.N, Np = ((N + 15) >> 4) * 16, Mp = ((M + 15) >> 4) * 16;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index over [Mp][Np]
    if (q >= (size_t)Mp * Np / 4) return;
    const int m = (int)(q / (Np / 4)), n = (int)(q % (Np / 4)) * 4;
    const size_t slab = (size_t)Mp * Np;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < S; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(d.part + (size_t)k * slab + (size_t)m * Np + n);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nn = min(n + r, N - 1);
        const float a = d.scale ? d.scale[nn] : 1.0f;
        const float b = d.shift ? d.shift[nn] : 0.0f;
        o[r] = (n + r < N) ? nd_act(a * o[r] + b, d.act) : 0.f;
    }
    if (d.out_packed == 2) {
        *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(d.out) + nd_pkh(m, n, N >> 5)) =
            f16x4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
    } else if (d.out_packed) {
        *reinterpret_cast<float4*>(d.out + nd_pk(m, n, N >> 4)) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (m < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < N) d.out[(size_t)m * N + n + r] = o[r];
    }
}

// ---- host helpers ---------------------------------------------------------------------------
// Compute units of the CURRENT device (256 on an MI355X in SPX mode, fewer under CPX partitioning), queried once per
// device.  Launch geometry is a pure function of (shape, CU count); 256 is assumed when no device is visible (host-only
// plan introspection in the build container).
static inline int nd_num_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}
// Row fragments (16 rows each) per workgroup pass: 1, 2 or 4 -- and FIVE where that saves a whole pass over the weights: 65..80 rows
// (the reference's default batch of 70, configs/chest_x_ray.yml:66: one pass of 80 rows instead of two of 64), 129..160, ...
// Measured against 4 everywhere on one box (profiles/r04_rows5_ab.txt): B = 70, mc = 1: 546 k -> 794 k step*img/s (fp16 operands
// 1.50 M -> 2.12 M); fp16 operands at 640 rows (mc = 20): 4.76 M -> 5.33 M.
static inline int nd_pick_mt(int M) {
    if (M <= 16) return 1;
    if (M <= 32) return 2;
    const int f = (M + 15) / 16;
    return (f + 4) / 5 < (f + 3) / 4 ? 5 : 4;
}
static inline bool nd_use_splitk(int K) { return K >= 16384; }

// err: what preparing the launch returned (the dynamic-LDS attribute of the kernel on the current device); the launchers below hand it
// back instead of launching, so it reaches the caller through nd_set_err like any other HIP error
struct SkinnyLaunch { void* fn; dim3 grid; dim3 block; int cps; int S; unsigned lds; hipError_t err; };   // lds: dynamic LDS bytes of the launch

// A kernel that asks for more than 64 KiB of dynamic LDS has to be told so before its first launch (graph kernel nodes included), and
// hipFuncSetAttribute is PER DEVICE: remembered as one bit per device and kernel (devices beyond 63: set every time).  One table for the
// whole library (inline function: one instance across translation units); failures are returned, never cached.
inline hipError_t nd_allow_dynamic_lds(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, unsigned long long> done;      // kernel -> devices that have the attribute
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    unsigned long long& mask = done[fn];
    if (dev >= 0 && dev < 64 && ((mask >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) mask |= 1ull << dev;
    return e;
}

// Geometry.  Measured (tools/ubench_skinny.hip, "sx" study, 5 x 67 MB at M = 32): weight stream alone 51 us, f32 MFMAs alone
// 40 us; together 80 us with 16 waves per CU, 66 us with 8, 57 us (min 51) with FOUR -- one wave per SIMD, each keeping two
// register stages so its own loads fly under its own MFMAs.  More waves per SIMD make the two overlap worse, not better.
// So: 4-wave workgroups, at most one per CU, every member's fragments spread evenly over its share of them (5 or 6 each at
// K = 5 members: 255 workgroups), U chunks per stage chosen to keep ~8-16 KiB per wave in flight.  Weights larger than what the 256 MiB Infinity Cache keeps
// across consecutive steps are streamed with nontemporal loads.
// MODE 2 (split-K): S k-slabs in grid.z so that (fragment groups) x (row groups) x S ~ 256 workgroups.
// The k_skinny family (4 row-fragment counts x 6 weight-fragment counts x 2 load policies x 2 operand types per MODE) is instantiated
// ONCE per MODE, each in a translation unit of its own (csrc/nd_skinny_m{0,1,2}.hip define ND_SKINNY_MODE before including this
// header); every other translation unit gets its launch plan -- kernel pointer included -- through these three functions.
SkinnyLaunch nd_skinny_launch_m0(int K, int N, int M, int nm, int half);
SkinnyLaunch nd_skinny_launch_m1(int K, int N, int M, int nm, int half);
SkinnyLaunch nd_skinny_launch_m2(int K, int N, int M, int nm, int half);
template <int MODE>
static inline SkinnyLaunch nd_skinny_launch(int K, int N, int M, int nm, int half = 0) {
    return MODE == 0 ? nd_skinny_launch_m0(K, N, M, nm, half) : MODE == 1 ? nd_skinny_launch_m1(K, N, M, nm, half) : nd_skinny_launch_m2(K, N, M, nm, half);
}

#ifdef ND_SKINNY_MODE
template <int MODE>
static inline SkinnyLaunch nd_skinny_launch_impl(int K, int N, int M, int nm, int half) {
    const int mt = nd_pick_mt(M);
    const int nfr = (N + 15) / 16, nch = half ? K / 32 : K / 16;
    const int mgroups = (M + 16 * mt - 1) / (16 * mt);
    const int nfmax = 6;                           // register budget (accumulators + two stages); LDS: see nd_skinny_red_bytes
    const int ncu = nd_num_cus();
    // workgroups per member: as many as keep the grid within one workgroup per CU, but no workgroup above nfmax fragments
    int wpm = ncu / nm;
    if (wpm < 1) wpm = 1;
    if (wpm > nfr) wpm = nfr;
    while ((nfr + wpm - 1) / wpm > nfmax) ++wpm;
    int S = 1, cps = nch;
    if (MODE == 2) {
        // split-K: k-slabs fill the CUs, so fragments per workgroup can stay high (each workgroup streams the x of its
        // slab once per NF weight fragments: at NF = 1 that is twice the weight bytes).  Pick (workgroups per member, S)
        // by cost = (1 + 0.3*MT/nf) / (fraction of the 256 CUs busy).
        double best = 1e30;
        int bw = wpm, bs = 1;
        for (int nfc = nfmax; nfc >= 1; --nfc) {
            const int w = (nfr + nfc - 1) / nfc;
            const int wgs = nm * w * mgroups;
            int sc = wgs >= ncu ? 1 : ncu / wgs;
            while (sc > 1 && nch / sc < 64) --sc;          // keep every slab >= 64 chunks deep
            const long tot = (long)wgs * sc;
            const double util = tot >= ncu ? (double)tot / ((double)ncu * ((tot + ncu - 1) / ncu)) : tot / (double)ncu;
            const double nfe = (double)nfr / w;
            const double cost = (1.0 + 0.3 * mt / nfe) / util;
            if (cost < best - 1e-9) { best = cost; bw = w; bs = sc; }
        }
        wpm = bw; S = bs;
        cps = (nch + S - 1) / S;
        S = (nch + cps - 1) / cps;
    }
    const int nf = (nfr + wpm - 1) / wpm;          // = base + (nfr % wpm != 0): the kernel's fragment slots
    const int gx = nm * wpm;
    const bool nt = (double)nm * N * (double)K * (half ? 2.0 : 4.0) > 160e6;
    SkinnyLaunch L{nullptr, dim3(gx, mgroups, S), dim3(256), cps, S, 0u, hipSuccess};
#define ND_SK(MTV, NFV, WV, UV)                                                                                     \
    (half ? (nt ? (void*)k_skinny<MTV, NFV, WV, UV, MODE, true, 1> : (void*)k_skinny<MTV, NFV, WV, UV, MODE, false, 1>) \
          : (nt ? (void*)k_skinny<MTV, NFV, WV, UV, MODE, true, 0> : (void*)k_skinny<MTV, NFV, WV, UV, MODE, false, 0>))
    if (mt == 5) {
        switch (nf) {
            case 1: L.fn = ND_SK(5, 1, 4, 2); break;
            case 2: L.fn = ND_SK(5, 2, 4, 2); break;
            case 3: L.fn = ND_SK(5, 3, 4, 2); break;
            case 4: L.fn = ND_SK(5, 4, 4, 2); break;
            case 5: L.fn = ND_SK(5, 5, 4, 2); break;
            default: L.fn = ND_SK(5, 6, 4, 2); break;
        }
    } else if (mt == 4) {
        switch (nf) {
            case 1: L.fn = ND_SK(4, 1, 4, 2); break;
            case 2: L.fn = ND_SK(4, 2, 4, 2); break;
            case 3: L.fn = ND_SK(4, 3, 4, 2); break;
            case 4: L.fn = ND_SK(4, 4, 4, 2); break;
            case 5: L.fn = ND_SK(4, 5, 4, 2); break;
            default: L.fn = ND_SK(4, 6, 4, 2); break;
        }
    } else if (mt == 2) {
        switch (nf) {
            case 1: L.fn = ND_SK(2, 1, 4, 4); break;
            case 2: L.fn = ND_SK(2, 2, 4, 4); break;
            case 3: L.fn = ND_SK(2, 3, 4, 2); break;
            case 4: L.fn = ND_SK(2, 4, 4, 2); break;
            case 5: L.fn = ND_SK(2, 5, 4, 2); break;
            default: L.fn = ND_SK(2, 6, 4, 2); break;
        }
    } else {
        switch (nf) {
            case 1: L.fn = ND_SK(1, 1, 4, 4); break;
            case 2: L.fn = ND_SK(1, 2, 4, 4); break;
            case 3: L.fn = ND_SK(1, 3, 4, 2); break;
            case 4: L.fn = ND_SK(1, 4, 4, 2); break;
            case 5: L.fn = ND_SK(1, 5, 4, 2); break;
            default: L.fn = ND_SK(1, 6, 4, 2); break;
        }
    }
#undef ND_SK
    if (nd_skinny_red_bytes(4, nf, mt) > ND_SKINNY_STATIC_LDS) {
        L.lds = (unsigned)nd_skinny_red_bytes(4, nf, mt);
        L.err = nd_allow_dynamic_lds(L.fn, L.lds);
    }
    return L;
}
#endif  // ND_SKINNY_MODE

static inline size_t nd_splitk_part_floats(int M, int K, int N, int nm = 1, int half = 0) {
    const SkinnyLaunch L = nd_skinny_launch<2>(K, N, M, nm, half);
    return (size_t)L.S * (size_t)(((M + 15) / 16) * 16) * (size_t)(((N + 15) / 16) * 16);
}

// descs: HOST copies of the nm members' descriptors (nm <= ND_INLINE_DESCS): passed by value
static inline hipError_t nd_launch_skinny_inline(const SkinnyLaunch& L, const SkinnyDesc* descs, int nm, int M, int t, hipStream_t st) {
    if (L.err != hipSuccess) return L.err;
    int cps = L.cps;
    SkinnyInline di{};
    for (int g = 0; g < nm && g < ND_INLINE_DESCS; ++g) di.d[g] = descs[g];
    const SkinnyDesc* table = nullptr;
    void* args[] = {&di, &table, &nm, &M, &t, &cps};
    return hipLaunchKernel(L.fn, L.grid, L.block, args, L.lds, st);
}

// d0: the descriptor of a single-member launch (table == nullptr, nm == 1), else ignored: table[0 .. nm) in device memory.
static inline hipError_t nd_launch_skinny(const SkinnyLaunch& L, SkinnyDesc d0, const SkinnyDesc* table, int nm, int M, int t,
                                          hipStream_t st) {
    if (L.err != hipSuccess) return L.err;
    int cps = L.cps;
    SkinnyInline di{};
    di.d[0] = d0;
    void* args[] = {&di, &table, &nm, &M, &t, &cps};
    return hipLaunchKernel(L.fn, L.grid, L.block, args, L.lds, st);
}
