// nd_persist.hip -- a whole p_sample_loop (diffusion/diffusion_utils.py:133-163) as ONE kernel launch.  gfx950 only.
//
// The hipGraph form of the loop (csrc/nd_sampler.hip) is 3T+1 kernel nodes: head, lin2 block, lin3 + lin4 block per step, all
// members in every launch.  This is the same loop with no launch boundary: each member's ~51 workgroups stay on their CUs for all T
// steps and meet only each other, at a per-member barrier after each of the three phases of a step (head | lin2 | lin3 + lin4);
// members may be started `skew` apart.  The arithmetic is k_skinny's and k_step_head's, instruction for instruction where it
// matters: the same fragment dealing (a workgroup owns the same 5 or 6 weight fragments of lin2 AND lin3 as in the graph form), the
// same per-wave interleaved K split, the same cross-wave sum order, the same reduction trees over the eps partials  =>  the same bits
// as the graph form (tests/test_gpu_sampler.py asserts equality on every state of every chain).
//
// MEASURED (round 6, EXPERIMENTS.md #14, profiles/r06_persistent_loop_experiment.txt): 1.08 x the graph form's time at K = 5, M = 32,
// F = 4096 -- the idea was that launch boundaries are chip-wide barriers at which no workgroup streams, and that a resident kernel
// could also keep part of its weights on chip.  The per-phase clocks say otherwise: a workgroup's loop is bound by its OWN exact-f32
// MFMAs (37 us for five fragments with the rest of the chip idle), the six-fragment workgroup of every member is the critical path
// and its 50 partners wait for it whether the wait is a kernel boundary or a barrier.  The form is therefore OPT-IN
// (nd_set_loop_form / ND_PERSIST=1); the default is the graph form.
//
// Hand-off between workgroups (MI355X_MICROARCH.md, inter-workgroup visibility; per-XCD L2s are not coherent, a CU's L1 is never
// refreshed): every byte another workgroup reads (h1, h2, the eps partials) is stored WRITE-THROUGH (`sc1`) and every load of it is
// an `sc1` load (L1 bypassed, L2-served); each storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at its own barrier, ONE
// lane adds to the member's arrival counter (agent-scope atomic), one lane polls that counter (sc1 loads, s_sleep between), the
// workgroup barrier again, then the loads.  No release / acquire fence (each costs 1.7 us and a whole-L2 write-back or an L1 flush).
// (The guide lists this hand-off form as MEASURED behaviour of gfx950 / ROCm 7.2, not as an architectural guarantee; here it is
// checked bit for bit against the graph form on every state of every chain over repeated replays -- one more reason the form is opt-in.)
// The barriers are sense-reversing and leave their counters clean, so a launch needs no reset; every spin is bounded: a wait that exceeds
// `spin_ticks` sets the sticky error word, and the workgroup (and, through that word, every other one) leaves the kernel --
// nd_persist_status reports it (and, asked to, clears the whole barrier block: an abandoned launch leaves counters behind).
// All workgroups of the grid must be resident at once: grid <= CU count, one workgroup per CU (by its LDS request), one process per
// device (the deployment the multi-GPU path has anyway; ND_PERSIST=0 selects the graph form for rehearsals that share a device).
#include "nd_persist.hpp"
#include <cstddef>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned* nd_gu32;

#ifdef ND_PERSIST_TIMING
// debug builds (tools/persist_times.py): per-workgroup time per phase, summed over the steps, in ticks of the 100 MHz clock:
// [workgroup][16] = {head: eps reduction + posterior, head: h1, wait 1, lin2 loop, lin2 epilogue, wait 2, lin3 loop, lin3 epilogue, wait 3}
__device__ long long* nd_persist_times = nullptr;
extern "C" int nd_debug_set_persist_times(void* dev_ptr) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nd_persist_times), &dev_ptr, sizeof(void*)) == hipSuccess ? 0 : -1;
}
#define PS_STAMP(k) do { const long long now_ = wall_clock64(); ps_acc[k] += now_ - ps_last; ps_last = now_; } while (0)
#else
#define PS_STAMP(k) do { } while (0)
#endif

#define PS_WAVES 4
#define PS_MT 2
#define PS_U 2

__device__ __forceinline__ unsigned ps_ld_sc1(const unsigned* p) { return __hip_atomic_load((nd_gu32)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ps_st_sc1(unsigned* p, unsigned v) { __hip_atomic_store((nd_gu32)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ps_ldf_sc1(const float* p) {
    return __hip_atomic_load((__attribute__((address_space(1))) float*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ps_stf_sc1(float* p, float v) {
    __hip_atomic_store((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// raw buffer over a whole activation matrix: base in SGPRs, byte offsets (< 2 GiB) per access; aux 16 = sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ps_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 ps_ld16_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    // (the whole vector is cast at once: __builtin_bit_cast of ONE element of an ext-vector reads element 0 whatever the index, clang 20)
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void ps_st16_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff, float4 f) {
    const f32x4 v = {f.x, f.y, f.z, f.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 16);
}

// Per-member barrier, sense-reversing, self-cleaning: `count` is 0 between barriers (the last arriver zeroes it BEFORE it publishes the
// next generation), `gen` only ever grows.  A workgroup reads the generation once at kernel start (it cannot move before this workgroup
// has arrived at the member's first barrier) and counts its own barriers from there, so a launch needs NO host-side reset: a memset
// node in front of the kernel node was tried first and is not ordered against the kernel on every replay path of ROCm 7.2's graph
// executor (replays enqueued back to back behind a device synchronise started with the previous launch's counter values: every
// barrier passed at once and all results were wrong -- tools/persist_check.py, profiles/r06_persistent_loop_experiment.txt).
// Returns false when the wait was abandoned (error word set, here or by another workgroup): the caller leaves the kernel.
__device__ __forceinline__ bool ps_member_barrier(unsigned* count, unsigned* gen, unsigned* errw, unsigned my_gen, unsigned wpm, int spin_ticks,
                                                  unsigned* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add((nd_gu32)count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1u;
        if (old + 1u == wpm) {                                // last arriver: clean the counter, then open the barrier
            ps_st_sc1(count, 0u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add((nd_gu32)gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const long long t0 = wall_clock64();
            while ((int)(ps_ld_sc1(gen) - my_gen) < 0) {      // (wrap-safe: generations are compared by their difference)
                __builtin_amdgcn_s_sleep(2);
                if (ps_ld_sc1(errw) != 0u) { ok = 0u; break; }
                if (wall_clock64() - t0 > (long long)spin_ticks) { ps_st_sc1(errw, 1u); ok = 0u; break; }
            }
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0u;
}

// LDS carve (floats), all in the dynamic region (16-byte aligned base)
template <int C, int NF>
struct PsLds {
    static constexpr int red = 0;                                   // [PS_WAVES][NF * PS_MT * 256]   cross-wave sum, then the reduced tiles
    static constexpr int ssc = red + PS_WAVES * NF * PS_MT * 256;   // [NF][16] scale of the running layer
    static constexpr int ssh = ssc + NF * 16;                       // [NF][16] shift
    static constexpr int pws = ssh + NF * 16;                       // [NF][C][16] lin4 rows of this workgroup's columns (constant)
    static constexpr int sy = pws + NF * C * 16;                    // [32 * C] y_t of every row of the member
    static constexpr int syh = sy + 32 * C;                         // [32 * C] yhat rows
    static constexpr int sym = syh + 32 * C;                        // [32 * C] y_T_mean rows
    static constexpr int sv = sym + 32 * C;                         // [PS_WAVES][64] partial sums of the eps reduction
    static constexpr int flag = sv + PS_WAVES * 64;                 // [4] barrier verdict
    static constexpr int total = flag + 4;
};

// Kernel arguments are re-read from the kernarg segment at the start of every phase through a pointer the compiler cannot see
// through: left alone it hoists all ~100 dwords of descriptors out of the step loop as loop invariants and then spills them to
// VGPR lanes around (and into) the streaming loops.
typedef const __attribute__((address_space(4))) char* ps_cbytes;
__device__ __forceinline__ ps_cbytes ps_launder(ps_cbytes p) { asm volatile("" : "+s"(p)); return p; }
#define PS_ARG(ka, type, member) nd_ldc<type>(ps_launder(ka) + offsetof(PersistArgs, member))
#define PS_ARG_G(ka, type, member, g) nd_ldc<type>(ps_launder(ka) + offsetof(PersistArgs, member) + (size_t)(g) * sizeof(type))

template <int C, int NF>
__global__ __launch_bounds__(PS_WAVES * 64) void k_persist_loop(PersistArgs args_by_value) {
    constexpr int MT = PS_MT, U = PS_U, WAVES = PS_WAVES;
    typedef PsLds<C, NF> L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ps_cbytes ka = (ps_cbytes)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x, lane = tid & 63, lane4 = lane * 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int F, M, B, T, nm, maxM, spin_ticks, fake_res;
    unsigned* bar;
    {
        const PersistScalars S = PS_ARG(ka, PersistScalars, s);
        F = S.F; M = S.M; B = S.B; T = S.T; nm = S.nm; maxM = S.maxM; spin_ticks = S.spin_ticks; bar = S.bar; fake_res = S.fake_resident;
        const int wpm0 = gridDim.x / nm, g0 = blockIdx.x / wpm0;
        if (g0 >= S.active) return;
        if (S.skew > 0 && g0 > 0) {                       // members start `skew` apart: their phases interleave instead of coinciding
            const long long t0 = wall_clock64(), dt = (long long)S.skew * g0;
            while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(8);
        }
    }
    const int nch = F >> 4, nfr = F >> 4, mtiles = (M + 15) >> 4, pairs = M * C;
    const int wpm = gridDim.x / nm;
    const int g = blockIdx.x / wpm, j = blockIdx.x - g * wpm;
    const int base = nfr / wpm, rem = nfr - base * wpm;
    const int nact = base + (j < rem ? 1 : 0);
    const int fi0 = j * base + min(j, rem);
    unsigned* const counter = bar + g * 32;               // arrival counter and generation word of this member, 64 bytes apart
    unsigned* const genw = bar + g * 32 + 16;
    unsigned* const errw = bar + ND_PERSIST_ERR_WORD;
    unsigned* const s_flag = reinterpret_cast<unsigned*>(smem + L::flag);
    if (tid == 0) *s_flag = ps_ld_sc1(genw);              // the member's generation as this launch finds it (see ps_member_barrier)
    __syncthreads();
    unsigned bar_gen = *s_flag;                           // generation that completes this workgroup's NEXT barrier, minus one
    __syncthreads();
#ifdef ND_PERSIST_TIMING
    long long ps_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ps_last = 0;
#endif

    // ---- once per launch: constants of this workgroup ----
    {
        const SkinnyDesc d3 = PS_ARG_G(ka, SkinnyDesc, l3, g);
        const StepIO io = PS_ARG(ka, StepIO, io);
        for (int e = tid; e < NF * C * 16; e += WAVES * 64) {
            const int f = e / (C * 16), c = (e / 16) % C, nl = e & 15;
            const int n = min(fi0 + f, nfr - 1) * 16 + nl;
            smem[L::pws + e] = (f < nact) ? nd_ldg(d3.pw + (size_t)c * F + n) : 0.f;
        }
        for (int p = tid; p < pairs; p += WAVES * 64) {
            const int m = p / C, c = p - m * C, b = m % B;
            smem[L::syh + p] = nd_ldg(io.yhat + g * io.yhat_ms + (size_t)b * C + c);
            smem[L::sym + p] = nd_ldg(io.ymean + g * io.ymean_ms + (size_t)b * C + c);
        }
    }
    __syncthreads();

    const int ngroups = nch / U;                          // the plan guarantees nch % U == 0
    const int ngw = ngroups > wave ? (ngroups - wave + WAVES - 1) / WAVES : 0;
    const int glast = ngroups > 0 ? ngroups - 1 : 0;
    float (*const red)[NF * MT * 256] = reinterpret_cast<float (*)[NF * MT * 256]>(smem + L::red);

    // A wait that was given up (the grid was not resident at once): the member's y_0 rows become NaNs, so no caller can mistake the
    // previous call's contents of the output buffer for a result; nd_persist_status reports the cause.
    auto poison = [&]() {
        if (j != 0) return;
        const StepIO io = PS_ARG(ka, StepIO, io);
        for (int p = tid; p < pairs; p += WAVES * 64) ND_GW(io.y0_out)[g * io.y0_ms + p] = __builtin_nanf("");
    };

    // ONE ConditionalLinear block of this workgroup's fragments (k_skinny's main loop and epilogue, MODE 0 / 1), entered through the
    // member's barrier: the first register stage of W and the epilogue's table entries do not depend on other workgroups and are
    // requested BEFORE the wait.  x: the frag16 activations the block reads (h1 / h2), out: h2 (MODE 0) or the eps partials (MODE 1).
    auto layer = [&](auto modec, int t) -> bool {
        constexpr int MODE = decltype(modec)::value;
        const float* wbase; const float* xbase; float* obase; int keep;
        float r_sc = 1.0f, r_sh = 0.0f;
        {
            const SkinnyDesc d = MODE == 0 ? PS_ARG_G(ka, SkinnyDesc, l2, g) : PS_ARG_G(ka, SkinnyDesc, l3, g);
            wbase = d.w; xbase = d.x; obase = MODE == 0 ? d.out : d.part; keep = d.keep;
            if (tid < NF * 16) {
                const int n = min(min(fi0 + (tid >> 4), nfr - 1) * 16 + (tid & 15), F - 1);
                r_sc = nd_ldg(d.scale + (size_t)t * F + n);
                r_sh = nd_ldg(d.shift + (size_t)t * F + n);
            }
        }
        const __amdgpu_buffer_rsrc_t rx = ps_rsrc(xbase);
        const float* wp[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) wp[f] = wbase + (size_t)min(fi0 + f, nfr - 1) * nch * 256;
        f32x4 acc[NF][MT];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bool alive = true;
        auto run = [&](auto ntc, auto fullc) {
            constexpr bool NTV = decltype(ntc)::value, FULL = decltype(fullc)::value;
            constexpr int NFA = FULL ? NF : (NF > 1 ? NF - 1 : 1);
            float4 wA[U][NFA], xA[U][MT], wB[U][NFA], xB[U][MT];
            auto LDW = [&](float4 (&w)[U][NFA], int grp) {
                const size_t go = (size_t)grp * U * 256;
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int f = 0; f < NFA; ++f) w[u][f] = nd_ld16<NTV>(wp[f] + go + u * 256 + lane4);
            };
            auto LDX = [&](float4 (&x)[U][MT], int grp) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        x[u][mt] = ps_ld16_sc1(rx, lane * 16, ((min(mt, mtiles - 1) * nch + grp * U + u) * 1024));
            };
            auto LD = [&](float4 (&w)[U][NFA], float4 (&x)[U][MT], int grp, int grpw) {
                // the graph form's issue order: per chunk, its weight fragments, then its activation fragments
                const size_t go = (size_t)grpw * U * 256;
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int f = 0; f < NFA; ++f) w[u][f] = nd_ld16<NTV>(wp[f] + go + u * 256 + lane4);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        x[u][mt] = ps_ld16_sc1(rx, lane * 16, ((min(mt, mtiles - 1) * nch + grp * U + u) * 1024));
                }
            };
            auto MMA = [&](const float4 (&w)[U][NFA], const float4 (&x)[U][MT]) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int f = 0; f < NFA; ++f)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                const float wv = jj == 0 ? w[u][f].x : jj == 1 ? w[u][f].y : jj == 2 ? w[u][f].z : w[u][f].w;
                                const float xv = jj == 0 ? x[u][mt].x : jj == 1 ? x[u][mt].y : jj == 2 ? x[u][mt].z : x[u][mt].w;
                                acc[f][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[f][mt], 0, 0, 0);
                            }
            };
            auto G = [&](int i) { return min(wave + i * WAVES, glast); };
            auto GW = [&](int i) { return i < fake_res ? min(wave, glast) : G(i); };   // (timing ablation: see PersistScalars::fake_resident)
            if (ngw > 0) LDW(wA, GW(0));                          // weights of the first stage: in flight across the barrier
            alive = ps_member_barrier(counter, genw, errw, ++bar_gen, (unsigned)wpm, spin_ticks, s_flag);
            if (!alive) return;
            PS_STAMP(MODE == 0 ? 2 : 5);
            if (ngw > 0) LDX(xA, G(0));
            constexpr int NL = U * (NFA + MT), NM = U * 4 * NFA * MT, MR = NM / NL > 0 ? NM / NL : 1;
#define PS_MIX()                                                                                     \
            _Pragma("unroll") for (int q_ = 0; q_ < NL; ++q_) {                                      \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                   \
                __builtin_amdgcn_sched_group_barrier(0x008, MR, 0);                                  \
            }                                                                                        \
            __builtin_amdgcn_sched_barrier(0);
            int i = 0;
            for (; i + 1 < ngw; i += 2) {
                LD(wB, xB, G(i + 1), GW(i + 1));
                MMA(wA, xA);
                PS_MIX()
                LD(wA, xA, G(i + 2), GW(i + 2));
                MMA(wB, xB);
                PS_MIX()
            }
#undef PS_MIX
            if (i < ngw) MMA(wA, xA);
        };
        const bool nt_here = !keep;
        if (nact == NF) {
            if (nt_here) run(std::true_type{}, std::true_type{});
            else run(std::false_type{}, std::true_type{});
        } else {
            if (nt_here) run(std::true_type{}, std::false_type{});
            else run(std::false_type{}, std::false_type{});
        }
        if (!alive) return false;
        PS_STAMP(MODE == 0 ? 3 : 6);
        // ---- epilogue: k_skinny's, fragment for fragment ----
        if (tid < NF * 16) { smem[L::ssc + tid] = r_sc; smem[L::ssh + tid] = r_sh; }
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][((f * MT + mt) * 4 + r) * 64 + lane] = acc[f][mt][r];
        __syncthreads();
        float* const R = red[0];
        constexpr int EPF = MT * 256 / (WAVES * 64);
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
                ev[f][jq] = smem[L::ssc + f * 16 + nl] * sum + smem[L::ssh + f * 16 + nl];
            }
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int jq = 0; jq < EPF; ++jq) ev[f][jq] = nd_softplus(ev[f][jq]);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int jq = 0; jq < EPF; ++jq) {
                const int q = f * MT * 256 + jq * WAVES * 64 + tid;
                R[q] = ev[f][jq];                          // (N = F is a multiple of 16: no column past N)
            }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t ro = ps_rsrc(obase);
        if (MODE == 0) {
            // h2: the 16x16 block (m-tile, fragment) is one contiguous 1 KiB of the frag16 output; write-through
            for (int e = tid; e < nact * MT * 64; e += WAVES * 64) {
                const int f = e / (MT * 64), mt = (e >> 6) % MT, Ln = e & 63;
                if (mt < mtiles) {
                    const float* tp = R + (f * MT + mt) * 256 + Ln;
                    ps_st16_sc1(ro, Ln * 16 + (mt * nfr + fi0 + f) * 1024, 0, make_float4(tp[0], tp[64], tp[128], tp[192]));
                }
            }
        } else {
            // projection onto lin4 (k_skinny MODE 1), one partial per (row, class, fragment), stored TILE-major [fragment][row][class]
            // so that a workgroup's partials are contiguous and the head's gather reads whole lines
            float* const pb = red[1];                      // planes 1.. of the sum buffer are free once the sum is taken
            for (int e = tid; e < nact * 16 * MT * C; e += WAVES * 64) {
                const int f = e / (16 * MT * C), ml = (e / C) % (16 * MT), c = e % C;
                if (ml < M) {
                    const float* tp = R + (f * MT + (ml >> 4)) * 256 + (ml & 15);
                    float sum = 0.f;
#pragma unroll
                    for (int nl = 0; nl < 16; ++nl) sum += smem[L::pws + (f * C + c) * 16 + nl] * tp[(nl & 3) * 64 + 16 * (nl >> 2)];
                    pb[f * pairs + ml * C + c] = sum;
                }
            }
            __syncthreads();
            if ((pairs & 3) == 0) {
                for (int e = tid; e < nact * pairs / 4; e += WAVES * 64)
                    ps_st16_sc1(ro, e * 16, fi0 * pairs * 4, *reinterpret_cast<const float4*>(pb + e * 4));
            } else {
                for (int e = tid; e < nact * pairs; e += WAVES * 64) ps_stf_sc1(obase + (size_t)fi0 * pairs + e, pb[e]);
            }
        }
        PS_STAMP(MODE == 0 ? 4 : 7);
        return true;
    };

    // ---- the T steps ----
#ifdef ND_PERSIST_TIMING
    ps_last = wall_clock64();
#endif
    for (int i = 0; i < T; ++i) {
        const int t = T - 1 - i, t_prev = t + 1;
        // head, part 1: y_t of every row of the member (redundantly in every workgroup: it needs all of them)
        {
            const StepIO io = PS_ARG(ka, StepIO, io);
            if (i == 0) {
                for (int p = tid; p < pairs; p += WAVES * 64) {
                    const int m = p / C, c = p - m * C;
                    smem[L::sy + p] = nd_ldg(io.noise + g * io.noise_ms + ((size_t)0 * M + m) * C + c) + smem[L::sym + p];
                }
            } else {
                const float* ppart = PS_ARG_G(ka, SkinnyDesc, l3, g).part;          // eps partials, TILE-major here: [nfr][pairs]
                const float* lin4_b = PS_ARG_G(ka, MemberDev, mem, g).lin4_b;
                const float al = nd_ldg(io.alphas + t_prev), s_t = nd_ldg(io.omabs + t_prev), s_tm1 = nd_ldg(io.omabs + t_prev - 1);
                for (int p0 = 0; p0 < pairs; p0 += 64) {
                    // nd_reduce_eps<256, C> of k_step_head, with lanes = (row, class) pairs and registers = partials: wave v sums
                    // partials 64 v .. 64 v + 63 in the shuffle tree's pairing (i, i + 32), (i, i + 16), ..., the four sums are then
                    // added in wave order
                    const int p = min(p0 + lane, pairs - 1);
                    float x[64];
#pragma unroll
                    for (int k = 0; k < 64; ++k) {
                        const int tl = 64 * wave + k;
                        const float v = ps_ldf_sc1(ppart + (size_t)min(tl, nfr - 1) * pairs + p);
                        x[k] = tl < nfr ? 0.f + v : 0.f;
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                        for (int k = 0; k < off; ++k) x[k] = x[k] + x[k + off];
                    smem[L::sv + wave * 64 + lane] = x[0];
                    __syncthreads();
                    if (tid < 64 && p0 + tid < pairs) {
                        const int pp = p0 + tid, m = pp / C, c = pp - m * C;
                        float tot = 0.f;
#pragma unroll
                        for (int w = 0; w < WAVES; ++w) tot += smem[L::sv + w * 64 + tid];
                        const float zz = nd_ldg(io.noise + g * io.noise_ms + ((size_t)i * M + m) * C + c);
                        smem[L::sy + pp] = nd_posterior(smem[L::sy + pp], smem[L::sym + pp], tot + nd_ldg(lin4_b + c), zz, al, s_t, s_tm1);
                    }
                    __syncthreads();
                }
            }
            __syncthreads();
            if (j == 0) {
                float* ybuf = PS_ARG_G(ka, MemberDev, mem, g).ybuf;
                for (int p = tid; p < pairs; p += WAVES * 64) {
                    const int m = p / C, c = p - m * C;
                    const float yv = smem[L::sy + p];
                    ND_GW(ybuf)[((size_t)(i & 1) * maxM + m) * C + c] = yv;
                    if (io.seq_out) ND_GW(io.seq_out)[g * io.seq_ms + ((size_t)i * M + m) * C + c] = yv;
                }
            }
        }
        PS_STAMP(0);
        // head, part 2: this workgroup's columns of h1 = softplus(A1[t] * (lin1.W [y_t, yhat]) + C1[t]) * xe  (k_step_head's element)
        {
            const MemberDev mb = PS_ARG_G(ka, MemberDev, mem, g);
            const __amdgpu_buffer_rsrc_t r_h1 = ps_rsrc(mb.h1);
            for (int e = tid; e < nact * MT * 64; e += WAVES * 64) {
                const int f = e / (MT * 64), mt = (e >> 6) % MT, Ln = e & 63;
                const int m = mt * 16 + (Ln & 15), n = (fi0 + f) * 16 + 4 * (Ln >> 4);
                if (m < M) {
                    const int b = m % B;
                    const float4 a4 = nd_ld16<false>(mb.A1 + (size_t)t * F + n), c4 = nd_ld16<false>(mb.C1 + (size_t)t * F + n);
                    const float4 xe = nd_ld16<false>(mb.xe + nd_pk(b, n, nch));
                    nd_gcf wrow = ND_GC(mb.lin1_w + (size_t)n * 2 * C);
                    float w1[4][2 * C];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int q = 0; q < 2 * C; ++q) w1[jj][q] = wrow[jj * 2 * C + q];
                    float yv[C], yh[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) { yv[c] = smem[L::sy + m * C + c]; yh[c] = smem[L::syh + m * C + c]; }
                    float4 h;
                    h.x = nd_head_element<C>(w1[0], yv, yh, a4.x, c4.x, xe.x);
                    h.y = nd_head_element<C>(w1[1], yv, yh, a4.y, c4.y, xe.y);
                    h.z = nd_head_element<C>(w1[2], yv, yh, a4.z, c4.z, xe.z);
                    h.w = nd_head_element<C>(w1[3], yv, yh, a4.w, c4.w, xe.w);
                    ps_st16_sc1(r_h1, Ln * 16 + (mt * nch + fi0 + f) * 1024, 0, h);
                }
            }
        }
        PS_STAMP(1);
        if (!layer(std::integral_constant<int, 0>{}, t)) { poison(); return; }
        if (!layer(std::integral_constant<int, 1>{}, t)) { poison(); return; }
        if (!ps_member_barrier(counter, genw, errw, ++bar_gen, (unsigned)wpm, spin_ticks, s_flag)) { poison(); return; }
        PS_STAMP(8);
    }
#ifdef ND_PERSIST_TIMING
    if (tid == 0 && nd_persist_times)
        for (int k = 0; k < 9; ++k) nd_persist_times[(size_t)blockIdx.x * 16 + k] = ps_acc[k];
#endif
    // ---- t = 0 behind the loop: y_0 (k_step_final: nd_reduce_eps<64, C> = four partials per lane added in order, then the tree) ----
    if (j == 0 && wave == 0) {
        const StepIO io = PS_ARG(ka, StepIO, io);
        const float* ppart = PS_ARG_G(ka, SkinnyDesc, l3, g).part;
        const float* lin4_b = PS_ARG_G(ka, MemberDev, mem, g).lin4_b;
        const float s0 = nd_ldg(io.omabs + 0);
        for (int p0 = 0; p0 < pairs; p0 += 64) {
            const int p = min(p0 + lane, pairs - 1), m = p / C, c = p - m * C;
            float x[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tl = k + 64 * q;
                    const float v = ps_ldf_sc1(ppart + (size_t)min(tl, nfr - 1) * pairs + p);
                    if (tl < nfr) s += v;
                }
                x[k] = s;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int k = 0; k < off; ++k) x[k] = x[k] + x[k + off];
            float tot = 0.f;
            tot += x[0];
            if (p0 + lane < pairs) {
                const float y0 = nd_y0_reparam(smem[L::sy + p], smem[L::sym + p], tot + nd_ldg(lin4_b + c), s0);
                ND_GW(io.y0_out)[g * io.y0_ms + (size_t)m * C + c] = y0;
                if (io.seq_out) ND_GW(io.seq_out)[g * io.seq_ms + ((size_t)T * M + m) * C + c] = y0;
            }
        }
    }
}

template <int NF>
static void* persist_fn(int C) {
    switch (C) {
        case 1: return (void*)k_persist_loop<1, NF>;
        case 2: return (void*)k_persist_loop<2, NF>;
        case 3: return (void*)k_persist_loop<3, NF>;
        case 4: return (void*)k_persist_loop<4, NF>;
        default: return nullptr;
    }
}
template <int NF>
static unsigned persist_lds(int C) {
    switch (C) {
        case 1: return PsLds<1, NF>::total * 4;
        case 2: return PsLds<2, NF>::total * 4;
        case 3: return PsLds<3, NF>::total * 4;
        default: return PsLds<4, NF>::total * 4;
    }
}

// The one-launch form runs where the graph form's blocks are k_skinny<2, NF in 3..6, 4, 2, MODE, NT = true> -- 17 .. 32 rows, fp32
// operands, descriptors by value, a launch that streams more than the Infinity Cache keeps (at fewer bytes a step is launch-latency
// bound and the barriers cost more than the boundaries they replace: tools/ubench_persist.hip) -- with the SAME fragment dealing.
PersistPlan nd_persist_plan(int F, int M, int nm, int C, int half) {
    PersistPlan p{false, nullptr, dim3(1), dim3(PS_WAVES * 64), 0u, hipSuccess, ""};
    if (half) { p.why = "fp16 operands"; return p; }
    if (M < 17 || M > 32) { p.why = "rows per member outside 17..32"; return p; }
    if (nm < 1 || nm > ND_INLINE_DESCS) { p.why = "member count"; return p; }
    if (C < 1 || C > 4) { p.why = "more than 4 classes"; return p; }
    if (F % 32 || F / 16 > 256 || F / 16 / PS_U < 2 * PS_WAVES) { p.why = "feature width"; return p; }
    const SkinnyLaunch L0 = nd_skinny_launch<0>(F, F, M, nm, 0);
    const int nfr = F / 16, wpm = (int)L0.grid.x / nm, nf = (nfr + wpm - 1) / wpm;
    if (L0.grid.y != 1 || (int)L0.grid.x > nd_num_cus()) { p.why = "grid"; return p; }
    if (!((double)nm * F * (double)F * 4.0 > 160e6)) { p.why = "launch-latency bound (weights fit the Infinity Cache)"; return p; }
    if (nf < 3 || nf > 6) { p.why = "fragment slots outside 3..6"; return p; }
    switch (nf) {
        case 3: p.fn = persist_fn<3>(C); p.lds = persist_lds<3>(C); break;
        case 4: p.fn = persist_fn<4>(C); p.lds = persist_lds<4>(C); break;
        case 5: p.fn = persist_fn<5>(C); p.lds = persist_lds<5>(C); break;
        default: p.fn = persist_fn<6>(C); p.lds = persist_lds<6>(C); break;
    }
    // one workgroup per CU, whatever the arrays add up to: two of these on a CU would still be correct (all resident), but the
    // streaming loop is built for one wave per SIMD
    if (p.lds < 84 * 1024) p.lds = 84 * 1024;
    p.grid = dim3(L0.grid.x);
    p.err = nd_allow_dynamic_lds(p.fn, p.lds);
    p.ok = true;
    return p;
}
