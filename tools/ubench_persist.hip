// ubench_persist.hip -- is ONE persistent launch per p_sample_loop (grid barriers between the layers) faster than the 3T+1-node
// hipGraph?  Measures the decisive quantities on the box itself, with a body shaped like the library's step blocks:
//   * 255/256 workgroups of 4 waves (one per SIMD), each streaming its share of a layer's weights with 1 KiB wave loads
//     (nontemporal, as k_skinny) under v_mfma_f32_16x16x4_f32 (8 per KiB, the real ratio at MT = 2), reading an L2-resident
//     activation block, writing ~10 KB of output per workgroup;
//   * a "step" = a small latency-bound head phase + two such layers (lin2, lin3), as nd_sampler.hip's step.
// Variants timed for G members (G * 64 MiB per layer):
//   graph     three kernel nodes per step, replayed from a hipGraph (what the library does)
//   persist   one launch for all steps; an XCD-hierarchical grid barrier (per-XCD arrival counter, the XCD's last arriver issues
//             the agent-scope release and arrives at the top counter, the last XCD publishes the generation; every workgroup
//             polls its XCD's generation word with sc1 loads and issues the agent-scope acquire) after every phase
//   persist+p the same with the first register stage of the NEXT layer's weights requested before the barrier wait
//   barrier   the barrier alone (empty phases)
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_persist tools/ubench_persist.hip ; run: ./ubench_persist [G=1] [steps=100] [reps=5]
// Every spin is bounded (a stuck barrier sets an error flag and the kernel drains), so a bug cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Bar {                       // every word on a 128-byte line of its own
    unsigned xcc_count[8][32];
    unsigned top[32];
    unsigned xcc_gen[8][32];
    unsigned members[8][32];       // workgroups that registered on each XCD (filled by k_census)
    unsigned err[32];
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// census: how many of the grid's workgroups run on each XCD (same grid and block size as the persistent kernel)
__global__ __launch_bounds__(256) void k_census(Bar* b) {
    if (threadIdx.x == 0) atomicAdd(&b->members[xcc_id() & 7][0], 1u);
}

// gen: this workgroup's barrier generation (1, 2, ...).  All waves must have drained their stores (s_waitcnt vmcnt(0)) before.
__device__ __forceinline__ void grid_barrier(Bar* b, unsigned gen, unsigned n_xcd_active) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id() & 7;
        const unsigned n_here = ld_sc1(&b->members[x][0]);
        const unsigned old = __hip_atomic_fetch_add(&b->xcc_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == n_here * gen) {                                  // this XCD's last arriver: publish the XCD's L2
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned o2 = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1 == n_xcd_active * gen)
                for (int k = 0; k < 8; ++k) st_sc1(&b->xcc_gen[k][0], gen);
        }
        long long t0 = wall_clock64();
        while (ld_sc1(&b->xcc_gen[x][0]) < gen) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ll) { st_sc1(&b->err[0], 1u); break; }       // ~2 s at 100 MHz: give up, never hang
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// flat per-MEMBER barrier (round 6): the G members' step chains are independent, so a barrier only has to join one member's workgroups.
// No census, no XCD hierarchy: every workgroup releases (FENCE: its own agent-scope release fence; the write-back covers its XCD's L2)
// and arrives at its member's counter; the last arriver publishes the member's generation word.
struct MBar {
    unsigned count[8][32];
    unsigned gen[8][32];
    unsigned err[32];
};
template <bool FENCE>
__device__ __forceinline__ void member_barrier(MBar* b, int g, unsigned gen, unsigned n_here) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const unsigned old = __hip_atomic_fetch_add(&b->count[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == n_here * gen) st_sc1(&b->gen[g][0], gen);
        long long t0 = wall_clock64();
        while (ld_sc1(&b->gen[g][0]) < gen) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ll) { st_sc1(&b->err[0], 1u); break; }
        }
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
}

struct P {
    const float* w[2];       // two layers' weights, G * 64 MiB each
    const float* x;          // activations (512 KB per member)
    float* out;              // per-workgroup outputs
    size_t kib_per_wave;     // weight KiB each wave streams per layer
    int G;
};

__device__ __forceinline__ f32x4 ld_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }

// one layer for this workgroup: stream kib_per_wave KiB per wave, 4 KiB in flight per wave per stage, 2 stages
template <bool PRE>
__device__ __forceinline__ void layer_body(const P& p, int layer, int wg, f32x4 (&pre)[4], bool have_pre, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* w = p.w[layer] + ((size_t)(wg * 4 + wave) * p.kib_per_wave) * 256 + lane * 4;
    const float* x = p.x + (size_t)(wg % p.G) * 131072 + wave * 32768 + lane * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 a[4], bq[4];
    const size_t n = p.kib_per_wave;           // multiple of 8
    if (PRE && have_pre) { for (int u = 0; u < 4; ++u) a[u] = pre[u]; }
    else { for (int u = 0; u < 4; ++u) a[u] = ld_nt(w + (size_t)u * 256); }
    for (size_t k = 0; k < n; k += 8) {
#pragma unroll
        for (int u = 0; u < 4; ++u) bq[u] = ld_nt(w + (k + 4 + u) * 256);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((k * 64) & 16383));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], xv[j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], xv[3 - j], acc, 0, 0, 0);
            }
        const size_t kn = k + 8 < n ? k + 8 : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = ld_nt(w + (kn + u) * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[u][j], xv[j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[u][j], xv[3 - j], acc, 0, 0, 0);
            }
    }
    // ~10 KB of output per workgroup (2.5 KB per wave), plain stores as the library's epilogue
    float* o = sink + ((size_t)wg * 4 + wave) * 640 + lane * 4;
    *reinterpret_cast<f32x4*>(o) = acc;
    *reinterpret_cast<f32x4*>(o + 256) = acc * 2.f;
    if (lane < 32) *reinterpret_cast<f32x4*>(o + 512) = acc * 3.f;
}

// head-like phase: every workgroup reads 2 KB written by OTHER workgroups in the previous phase and writes 2 KB
__device__ __forceinline__ void head_body(const P& p, int wg, int nwg, int it) {
    const float* src = p.out + ((size_t)((wg * 7 + it) % nwg) * 4) * 640;
    const float v = src[threadIdx.x] + src[256 + threadIdx.x];
    p.out[(size_t)nwg * 2560 + (size_t)wg * 256 + threadIdx.x] = v * 0.5f;
}

__global__ __launch_bounds__(256) void k_layer(P p, int layer) { f32x4 d[4]; layer_body<false>(p, layer, blockIdx.x, d, false, p.out); }
__global__ __launch_bounds__(256) void k_head(P p, int nwg, int it) { head_body(p, blockIdx.x, nwg, it); }

template <bool PRE, bool WORK>
__global__ __launch_bounds__(256) void k_persist(P p, Bar* b, int steps, unsigned n_xcd_active, unsigned gen0) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned gen = gen0;
    f32x4 pre[4];
    for (int it = 0; it < steps; ++it) {
        if (WORK) head_body(p, wg, nwg, it);
        if (PRE && WORK) for (int u = 0; u < 4; ++u) pre[u] = ld_nt(p.w[0] + ((size_t)(wg * 4 + wave) * p.kib_per_wave + u) * 256 + lane * 4);
        grid_barrier(b, ++gen, n_xcd_active);
        if (WORK) layer_body<PRE>(p, 0, wg, pre, true, p.out);
        if (PRE && WORK) for (int u = 0; u < 4; ++u) pre[u] = ld_nt(p.w[1] + ((size_t)(wg * 4 + wave) * p.kib_per_wave + u) * 256 + lane * 4);
        grid_barrier(b, ++gen, n_xcd_active);
        if (WORK) layer_body<PRE>(p, 1, wg, pre, true, p.out);
        grid_barrier(b, ++gen, n_xcd_active);
    }
}

template <bool PRE, bool WORK, bool FENCE>
__global__ __launch_bounds__(256) void k_persist_m(P p, MBar* b, int steps, unsigned gen0) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpm = nwg / p.G, g = wg / wpm;
    unsigned gen = gen0;
    f32x4 pre[4];
    for (int it = 0; it < steps; ++it) {
        if (WORK) head_body(p, wg, nwg, it);
        if (PRE && WORK) for (int u = 0; u < 4; ++u) pre[u] = ld_nt(p.w[0] + ((size_t)(wg * 4 + wave) * p.kib_per_wave + u) * 256 + lane * 4);
        member_barrier<FENCE>(b, g, ++gen, wpm);
        if (WORK) layer_body<PRE>(p, 0, wg, pre, true, p.out);
        if (PRE && WORK) for (int u = 0; u < 4; ++u) pre[u] = ld_nt(p.w[1] + ((size_t)(wg * 4 + wave) * p.kib_per_wave + u) * 256 + lane * 4);
        member_barrier<FENCE>(b, g, ++gen, wpm);
        if (WORK) layer_body<PRE>(p, 1, wg, pre, true, p.out);
        member_barrier<FENCE>(b, g, ++gen, wpm);
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 1, steps = argc > 2 ? atoi(argv[2]) : 100, reps = argc > 3 ? atoi(argv[3]) : 5;
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int nwg = (ncu / G) * G;                              // 256 at G = 1, 255 at G = 5: as nd_skinny_launch
    const size_t layer_bytes = (size_t)G * 4096 * 4096 * 4;
    size_t kib_per_wave = layer_bytes / 1024 / ((size_t)nwg * 4);
    kib_per_wave -= kib_per_wave % 8;
    printf("G=%d members: %d workgroups x 4 waves, %zu KiB per wave per layer (%.1f MB per layer), %d steps\n", G, nwg, kib_per_wave,
           (double)kib_per_wave * 1024 * nwg * 4 / 1e6, steps);
    float *w0, *w1, *x, *out;
    Bar* bar;
    CK(hipMalloc(&w0, layer_bytes + (1 << 20))); CK(hipMalloc(&w1, layer_bytes + (1 << 20)));
    CK(hipMalloc(&x, (size_t)G * 131072 * 4)); CK(hipMalloc(&out, (size_t)nwg * 2560 * 4 + (size_t)nwg * 256 * 4));
    CK(hipMalloc(&bar, sizeof(Bar)));
    CK(hipMemset(w0, 0, layer_bytes + (1 << 20))); CK(hipMemset(w1, 0, layer_bytes + (1 << 20)));
    CK(hipMemset(x, 0, (size_t)G * 131072 * 4)); CK(hipMemset(out, 0, (size_t)nwg * 2560 * 4 + (size_t)nwg * 256 * 4));
    CK(hipMemset(bar, 0, sizeof(Bar)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    P p{{w0, w1}, x, out, kib_per_wave, G};
    hipLaunchKernelGGL(k_census, dim3(nwg), dim3(256), 0, st, bar);
    CK(hipStreamSynchronize(st));
    Bar hb;
    CK(hipMemcpy(&hb, bar, sizeof hb, hipMemcpyDeviceToHost));
    unsigned n_xcd_active = 0, tot = 0;
    printf("workgroups per XCD:");
    for (int k = 0; k < 8; ++k) { printf(" %u", hb.members[k][0]); n_xcd_active += hb.members[k][0] > 0; tot += hb.members[k][0]; }
    printf("  (%u XCDs, %u workgroups)\n", n_xcd_active, tot);
    if ((int)tot != nwg) { printf("census mismatch\n"); return 1; }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& fn) {
        fn(); CK(hipStreamSynchronize(st));
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%-28s %8.3f ms best, %8.3f ms mean  = %7.2f us per step\n", name, best, sum / reps, 1e3 * best / steps);
        return best;
    };
    // --- graph: head, lin2, lin3 kernel nodes per step ---
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int it = 0; it < steps; ++it) {
        hipLaunchKernelGGL(k_head, dim3(nwg), dim3(256), 0, st, p, nwg, it);
        hipLaunchKernelGGL(k_layer, dim3(nwg), dim3(256), 0, st, p, 0);
        hipLaunchKernelGGL(k_layer, dim3(nwg), dim3(256), 0, st, p, 1);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const float t_graph = timeit("graph (3 nodes per step)", [&] { CK(hipGraphLaunch(ge, st)); });
    // layers alone, back to back (no head): the streaming floor of this body
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int it = 0; it < steps; ++it) { hipLaunchKernelGGL(k_layer, dim3(nwg), dim3(256), 0, st, p, 0); hipLaunchKernelGGL(k_layer, dim3(nwg), dim3(256), 0, st, p, 1); }
    CK(hipStreamEndCapture(st, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    timeit("graph, the two layers only", [&] { CK(hipGraphLaunch(ge2, st)); });
    // --- persistent ---
    unsigned gen0 = 0;
    auto run_persist = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, st, p, bar, steps, n_xcd_active, gen0);
        gen0 += 3u * (unsigned)steps;
    };
    const float t_bar = timeit("persistent: barriers only", [&] { run_persist(k_persist<false, false>); });
    const float t_p = timeit("persistent", [&] { run_persist(k_persist<false, true>); });
    const float t_pp = timeit("persistent + prefetch", [&] { run_persist(k_persist<true, true>); });
    // --- flat per-member barriers ---
    MBar* mbar; CK(hipMalloc(&mbar, sizeof(MBar))); CK(hipMemset(mbar, 0, sizeof(MBar)));
    unsigned mgen0 = 0;
    auto run_m = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, st, p, mbar, steps, mgen0);
        mgen0 += 3u * (unsigned)steps;
    };
    const float t_mbar = timeit("member barriers only (fenced)", [&] { run_m(k_persist_m<false, false, true>); });
    const float t_mbar0 = timeit("member barriers only (no fence)", [&] { run_m(k_persist_m<false, false, false>); });
    const float t_mp = timeit("persistent, member barriers", [&] { run_m(k_persist_m<false, true, true>); });
    const float t_mpp = timeit("persistent + prefetch, member", [&] { run_m(k_persist_m<true, true, true>); });
    const float t_mpp0 = timeit("  the same, fences removed", [&] { run_m(k_persist_m<true, true, false>); });
    MBar hm; CK(hipMemcpy(&hm, mbar, sizeof hm, hipMemcpyDeviceToHost));
    printf("member barrier: %.2f us each fenced, %.2f without fences; persistent / graph = %.3f, with prefetch %.3f (no fences: %.3f); error flag %u\n",
           1e3 * t_mbar / (3.0 * steps), 1e3 * t_mbar0 / (3.0 * steps), t_mp / t_graph, t_mpp / t_graph, t_mpp0 / t_graph, hm.err[0]);
    CK(hipMemcpy(&hb, bar, sizeof hb, hipMemcpyDeviceToHost));
    printf("barrier: %.2f us each; persistent / graph = %.3f, with prefetch %.3f; barrier error flag %u\n", 1e3 * t_bar / (3.0 * steps), t_p / t_graph,
           t_pp / t_graph, hb.err[0]);
    return (hb.err[0] || hm.err[0]) ? 2 : 0;
}
