// nd_ops.hip -- standalone operators: skinny Linear (mapping MLP), row softmax, ensemble aggregation.
// gfx950 only.
#include "nd_common.hpp"
#include "nd_cond_gemm.hpp"
#include "../../include/nested_diffusion.h"

int nd_set_err(int code, const char* fmt, ...);
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return nd_set_err(ND_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ---- packing ---------------------------------------------------------------------------------
static int bad_dtype(int dtype) { return dtype != ND_DTYPE_F32 && dtype != ND_DTYPE_F16; }
static int kmul(int dtype) { return dtype == ND_DTYPE_F16 ? 32 : 16; }

extern "C" size_t nd_packed_bytes(int R, int K, int dtype) {
    if (bad_dtype(dtype) || R < 1 || K < kmul(dtype) || (K % kmul(dtype))) return 0;
    return nd_packed_bytes_dt(R, K, dtype == ND_DTYPE_F16);
}

extern "C" int nd_pack_rows(const float* src, void* dst, int R, int K, int dtype, void* stream) {
    if (!src || !dst) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (bad_dtype(dtype)) return nd_set_err(ND_ERR_ARG, "unknown dtype %d", dtype);
    if (R < 1 || K < kmul(dtype) || (K % kmul(dtype)))
        return nd_set_err(ND_ERR_ARG, "need R >= 1 and K a positive multiple of %d (K=%d)", kmul(dtype), K);
    const int half = dtype == ND_DTYPE_F16;
    const size_t n16 = nd_packed_bytes_dt(R, K, half) / 16, want = (n16 + 255) / 256;
    const dim3 grid((unsigned)(want > 8192 ? 8192 : want));
    if (half) hipLaunchKernelGGL(k_pack_rows_h, grid, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, R, K);
    else hipLaunchKernelGGL(k_pack_rows, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, R, K);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

// ---- nd_linear: mapping/models/mlp.py:25-28 ---------------------------------------------------
extern "C" size_t nd_linear_workspace_bytes(int M, int K, int N, int dtype) {
    if (bad_dtype(dtype) || M < 1 || K < kmul(dtype) || (K % kmul(dtype)) || N < 1) return 0;
    const int half = dtype == ND_DTYPE_F16;
    size_t bytes = nd_packed_bytes_dt(M, K, half) + 256;
    if (nd_use_splitk(K)) bytes += nd_splitk_part_floats(M, K, N, 1, half) * sizeof(float);
    else bytes += nd_cond_gemm_plan(K, N, M, 1, half).ws_bytes;      // k-slab accumulators of the large-M kernel's split tail
    return bytes + 256;
}

// One Linear on an ALREADY PACKED input: out = act(scale * (x . W^T) + shift); out_packed 0 row-major fp32, 1 frag16, 2 frag32h (the
// layout the next streaming layer reads).  part: split-K partial sums / k-slab accumulators (nd_linear_workspace_bytes minus the
// packed-x share).
static int linear_packed(const float* xpk, const float* wf, const float* scale, const float* shift, float* out, int M, int K, int N, int act,
                         int half, int out_packed, float* part, hipStream_t st) {
    if (nd_use_splitk(K)) {
        const SkinnyLaunch L = nd_skinny_launch<2>(K, N, M, 1, half);
        SkinnyDesc sd{xpk, wf, nullptr, nullptr, nullptr, nullptr, part, K, N, 0, ND_ACT_NONE, 0};
        HIP_CHECK(nd_launch_skinny(L, sd, nullptr, 1, M, 0, st));
        SplitKEpiDesc se{part, scale, shift, out, N, L.S, act, out_packed};
        const size_t q = (size_t)(((M + 15) / 16) * 16) * (((N + 15) / 16) * 16) / 4;
        hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, se, (const SplitKEpiDesc*)nullptr, M, L.S);
    } else {
        SkinnyDesc d{xpk, wf, scale, shift, out, nullptr, nullptr, K, N, 0, act, out_packed};
        const CondGemmPlan tp = nd_cond_gemm_plan(K, N, M, 1, half);
        if (tp.use_tile) HIP_CHECK(nd_launch_cond_gemm(0, tp, d, nullptr, M, 0, part, st));      // more than 128 rows: LDS-tiled
        else HIP_CHECK(nd_launch_skinny(nd_skinny_launch<0>(K, N, M, 1, half), d, nullptr, 1, M, 0, st));
    }
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

extern "C" int nd_linear(const float* x, const void* wpk, const float* scale, const float* shift, float* out, int M, int K, int N,
                         int act, int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !wpk || !out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (bad_dtype(dtype)) return nd_set_err(ND_ERR_ARG, "unknown dtype %d", dtype);
    if (M < 1 || N < 1 || K < kmul(dtype) || (K % kmul(dtype)))
        return nd_set_err(ND_ERR_ARG, "need M,N >= 1 and K a positive multiple of %d (K=%d)", kmul(dtype), K);
    if (act < 0 || act > 3) return nd_set_err(ND_ERR_ARG, "unknown activation %d", act);
    const size_t need = nd_linear_workspace_bytes(M, K, N, dtype);
    if (!ws || ws_bytes < need) return nd_set_err(ND_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, need);
    const int half = dtype == ND_DTYPE_F16;
    float* xpk = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* part = (float*)((char*)xpk + ((nd_packed_bytes_dt(M, K, half) + 255) & ~(size_t)255));
    int rc = nd_pack_rows(x, xpk, M, K, dtype, stream);
    if (rc != ND_OK) return rc;
    return linear_packed(xpk, (const float*)wpk, scale, shift, out, M, K, N, act, half, 0, part, (hipStream_t)stream);
}

// mapping/models/mlp.py:23-29 as ONE sequence on packed activations (library-internal; nd_conditioner.hip): the input is packed
// once, every hidden layer's epilogue writes its output in the fragment order the next layer streams (the same values nd_linear's
// row-major output + re-pack would give, fp16 rounding included), the last layer writes row-major logits.
//   dims[5] = {in, w1, w2, w3, classes}; hid[l]: >= 16*ceil(M/16) * dims[l+1] floats; ws: >= the largest nd_linear_workspace_bytes of
//   the four layers.
int nd_mlp_chain(const float* x, const void* const* wpk, const float* const* bias, const int* dims, float* const* hid, float* logits,
                 int M, int dtype, void* ws, size_t ws_bytes, void* stream) {
    const int half = dtype == ND_DTYPE_F16, opk = half ? 2 : 1;
    for (int l = 0; l < 4; ++l)
        if (ws_bytes < nd_linear_workspace_bytes(M, dims[l], dims[l + 1], dtype))
            return nd_set_err(ND_ERR_ARG, "mlp chain workspace too small for layer %d", l + 1);
    float* xpk = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* part = (float*)((char*)xpk + ((nd_packed_bytes_dt(M, dims[0], half) + 255) & ~(size_t)255));
    int rc = nd_pack_rows(x, xpk, M, dims[0], dtype, stream);
    if (rc != ND_OK) return rc;
    const float* in = xpk;
    for (int l = 0; l < 4; ++l) {
        float* out = l < 3 ? hid[l] : logits;
        // layers 2..4 read a packed activation of the previous epilogue; their split-K / k-slab scratch can start at the workspace base
        rc = linear_packed(in, (const float*)wpk[l], nullptr, bias[l], out, M, dims[l], dims[l + 1], l < 3 ? ND_ACT_RELU : ND_ACT_NONE, half,
                           l < 3 ? opk : 0, l == 0 ? part : xpk, (hipStream_t)stream);
        if (rc != ND_OK) return rc;
        in = out;
    }
    return ND_OK;
}

// The same Classifier.forward for SEVERAL mapping MLPs whose first layers have already run (nd_mlp_chain_first): layers 2..4 of all nm
// members as THREE launches instead of 3 * nm -- the members' weights differ, their shapes do not, so one k_skinny launch streams the
// nm weight matrices side by side exactly as a step block streams the K noise estimators' (descriptors by value).  At config dims a
// single member's layer is 33 MB / 1 MB / 1 KB of weights: launch-latency bound on its own (11 us each, 15 launches per batch).
// Values are those of the one-member launches up to the dealing of k-chunks to waves (fixed per geometry: reproducible).
// batchable: every one of the three layers is a plain streaming launch (no split-K, no LDS-tiled form) and nm fits the inline table.
bool nd_mlp_tail_batchable(const int* dims, int M, int nm, int dtype) {
    const int half = dtype == ND_DTYPE_F16;
    if (nm < 2 || nm > ND_INLINE_DESCS) return false;
    for (int l = 1; l < 4; ++l)
        if (nd_use_splitk(dims[l]) || nd_cond_gemm_plan(dims[l], dims[l + 1], M, nm, half).use_tile || nd_cond_gemm_plan(dims[l], dims[l + 1], M, 1, half).use_tile)
            return false;
    return true;
}

// layer 1 of one mapping MLP (the 150528-wide split-K stream): hid0 = relu(x W1^T + b1), written packed for layer 2
int nd_mlp_chain_first(const float* x, const void* w1pk, const float* bias1, const int* dims, float* hid0, int M, int dtype, void* ws,
                       size_t ws_bytes, void* stream) {
    const int half = dtype == ND_DTYPE_F16, opk = half ? 2 : 1;
    if (ws_bytes < nd_linear_workspace_bytes(M, dims[0], dims[1], dtype)) return nd_set_err(ND_ERR_ARG, "mlp chain workspace too small for layer 1");
    float* xpk = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* part = (float*)((char*)xpk + ((nd_packed_bytes_dt(M, dims[0], half) + 255) & ~(size_t)255));
    int rc = nd_pack_rows(x, xpk, M, dims[0], dtype, stream);
    if (rc != ND_OK) return rc;
    return linear_packed(xpk, (const float*)w1pk, nullptr, bias1, hid0, M, dims[0], dims[1], ND_ACT_RELU, half, opk, part, (hipStream_t)stream);
}

// layers 2..4 of nm members: hid[l][k] = member k's activation after layer l+1 (packed), logits + k * logits_stride = its output
int nd_mlp_chain_tail(int nm, const nd_mlp_weights* w, const int* dims, float* const* hid0, float* const* hid1, float* const* hid2, float* logits,
                      size_t logits_stride, int M, int dtype, void* stream) {
    const int half = dtype == ND_DTYPE_F16, opk = half ? 2 : 1;
    if (!nd_mlp_tail_batchable(dims, M, nm, dtype)) return nd_set_err(ND_ERR_ARG, "mlp tail of %d members at %d rows is not batchable", nm, M);
    float* const* in[3] = {hid0, hid1, hid2};
    float* const* outp[2] = {hid1, hid2};
    for (int l = 1; l < 4; ++l) {
        SkinnyDesc d[ND_INLINE_DESCS];
        for (int k = 0; k < nm; ++k)
            d[k] = SkinnyDesc{in[l - 1][k], (const float*)w[k].w_packed[l], nullptr, w[k].bias[l], l < 3 ? outp[l - 1][k] : logits + (size_t)k * logits_stride,
                              nullptr, nullptr, dims[l], dims[l + 1], 0, l < 3 ? ND_ACT_RELU : ND_ACT_NONE, l < 3 ? opk : 0, 0};
        HIP_CHECK(nd_launch_skinny_inline(nd_skinny_launch<0>(dims[l], dims[l + 1], M, nm, half), d, nm, M, 0, (hipStream_t)stream));
    }
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

// ---- launch-plan introspection (host only; no GPU needed) -------------------------------------
extern "C" int nd_skinny_plan(int K, int N, int M, int n_members, int dtype, int mode, int* out6) {
    if (!out6) return nd_set_err(ND_ERR_ARG, "out6 is NULL");
    if (bad_dtype(dtype) || M < 1 || N < 1 || n_members < 1 || K < kmul(dtype) || (K % kmul(dtype)) || mode < 0 || mode > 2)
        return nd_set_err(ND_ERR_ARG, "bad shape / dtype / mode");
    const int half = dtype == ND_DTYPE_F16;
    const SkinnyLaunch L = mode == 0 ? nd_skinny_launch<0>(K, N, M, n_members, half)
                         : mode == 1 ? nd_skinny_launch<1>(K, N, M, n_members, half) : nd_skinny_launch<2>(K, N, M, n_members, half);
    const int nfr = (N + 15) / 16, wpm = (int)L.grid.x / n_members;
    out6[0] = (int)L.grid.x; out6[1] = (int)L.grid.y; out6[2] = (int)L.grid.z;
    out6[3] = (nfr + wpm - 1) / wpm;        // fragment slots per workgroup (the kernel's NF)
    out6[4] = L.cps;                        // k-chunks per slab
    out6[5] = (int)L.block.x;
    return ND_OK;
}

// Row fragments (16 rows each) a k_skinny workgroup keeps per pass over the weights at M rows: nd_pick_mt, the kernel's MT.
extern "C" int nd_skinny_row_fragments(int M) {
    if (M < 1) return nd_set_err(ND_ERR_ARG, "M must be >= 1");
    return nd_pick_mt(M);
}

// Which kernel a ConditionalLinear block of the sampler (K = N = F) runs at M = B*mc rows, and its tile plan.
extern "C" int nd_step_plan(int F, int M, int n_members, int dtype, int* out8) {
    if (!out8) return nd_set_err(ND_ERR_ARG, "out8 is NULL");
    if (bad_dtype(dtype) || M < 1 || n_members < 1 || F < kmul(dtype) || (F % kmul(dtype))) return nd_set_err(ND_ERR_ARG, "bad shape / dtype");
    const CondGemmPlan p = nd_cond_gemm_plan(F, F, M, n_members, dtype == ND_DTYPE_F16);
    out8[0] = p.use_tile;                              // 0: k_skinny (weight streaming), 1: k_cond_gemm (LDS-tiled)
    out8[1] = p.n_full + p.rem * p.split;              // workgroups of the tiled launch
    out8[2] = p.n_full; out8[3] = p.rem; out8[4] = p.split;
    out8[5] = p.use_tile ? p.ntl : (F + 15) / 16;      // partial sums per (row, class) left for the step head
    out8[6] = p.TM; out8[7] = p.TN;
    return ND_OK;
}

// ---- softmax over the class dim (classification_train_separately.py:755-758) -----------------
__global__ void k_softmax_rows(const float* __restrict__ x, float* __restrict__ out, int rows, int C) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* p = x + (size_t)r * C;
    float mx = p[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(p[c] - mx);
    for (int c = 0; c < C; ++c) out[(size_t)r * C + c] = expf(p[c] - mx) / s;
}

extern "C" int nd_softmax_rows(const float* x, float* out, int rows, int C, void* stream) {
    if (!x || !out || rows < 1 || C < 1) return nd_set_err(ND_ERR_ARG, "bad softmax arguments");
    hipLaunchKernelGGL(k_softmax_rows, dim3((rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, x, out, rows, C);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

// ---- aggregation (classification_train_separately.py:51-68, 392-398, 425-447) ------------------
// One thread per image b: loops the S samples in order (member-major then trial), so the mean is a
// fixed-order sum.  convert_to_prob: softmax(-(y-1)^2 / temperature); vote: mode of argmax over the
// raw y_0 (first maximum = smallest label on ties, as torch.argmax / torch.unique+counts.argmax).
#define ND_AGG_MAX_C 16
__global__ void k_aggregate(const float* __restrict__ samples, float* __restrict__ prob, long long* __restrict__ vote,
                            float* __restrict__ probs, int S, int B, int C, float temperature) {
#pragma clang fp contract(off)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float acc[ND_AGG_MAX_C];
    int cnt[ND_AGG_MAX_C];
    for (int c = 0; c < C; ++c) { acc[c] = 0.f; cnt[c] = 0; }
    for (int s = 0; s < S; ++s) {
        const float* y = samples + ((size_t)s * B + b) * C;
        float lg[ND_AGG_MAX_C];
        int am = 0;
        float best = y[0];
        for (int c = 0; c < C; ++c) {
            const float d = y[c] - 1.0f;
            lg[c] = d * d * (-1.0f) / temperature;
            if (y[c] > best) { best = y[c]; am = c; }
        }
        cnt[am]++;
        float mx = lg[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, lg[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) { lg[c] = expf(lg[c] - mx); sum += lg[c]; }
        for (int c = 0; c < C; ++c) {
            const float p = lg[c] / sum;
            acc[c] += p;
            if (probs) probs[((size_t)s * B + b) * C + c] = p;
        }
    }
    int vm = 0;
    for (int c = 1; c < C; ++c) if (cnt[c] > cnt[vm]) vm = c;
    vote[b] = vm;
    for (int c = 0; c < C; ++c) prob[(size_t)b * C + c] = acc[c] / (float)S;
}

extern "C" int nd_aggregate(const float* samples, float* prob_out, int64_t* vote_out, float* probs_out, int S, int B, int C,
                            float temperature, void* stream) {
    if (!samples || !prob_out || !vote_out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (S < 1 || B < 1 || C < 1 || C > ND_AGG_MAX_C) return nd_set_err(ND_ERR_ARG, "S,B >= 1 and 1 <= C <= %d required", ND_AGG_MAX_C);
    if (!(temperature > 0.f)) return nd_set_err(ND_ERR_ARG, "temperature must be > 0");
    hipLaunchKernelGGL(k_aggregate, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, samples, prob_out, (long long*)vote_out,
                       probs_out, S, B, C, temperature);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}


// ---- reporting tail (classification_train_separately.py:102-174, 413-423, 801-815) ---------------
// One wave per (image, class): the S per-sample values go to LDS; every lane ranks its elements by counting
// (stable: ties broken by index), the four order statistics needed by the two interpolated quantiles are
// picked by rank.  Variance: two-pass (mean, then centred squares), unbiased.
#define ND_STATS_MAXS 4096
__global__ __launch_bounds__(64) void k_sample_stats(const float* __restrict__ probs, float* __restrict__ piw, float* __restrict__ var,
                                                     int S, int B, int C, float q_lo, float q_hi) {
#pragma clang fp contract(off)
    const int bc = blockIdx.x, lane = threadIdx.x;
    __shared__ float v[ND_STATS_MAXS];
    __shared__ float pick[4];
    for (int s = lane; s < S; s += 64) v[s] = probs[(size_t)s * B * C + bc];
    __syncthreads();
    const float r_lo = q_lo * (float)(S - 1), r_hi = q_hi * (float)(S - 1);
    const int k0 = (int)floorf(r_lo), k2 = (int)floorf(r_hi);
    const int k1 = min(k0 + 1, S - 1), k3 = min(k2 + 1, S - 1);
    float sum = 0.f;
    for (int i = lane; i < S; i += 64) {
        const float vi = v[i];
        sum += vi;
        int rank = 0;
        for (int j = 0; j < S; ++j) rank += (v[j] < vi) || (v[j] == vi && j < i);
        if (rank == k0) pick[0] = vi;
        if (rank == k1) pick[1] = vi;
        if (rank == k2) pick[2] = vi;
        if (rank == k3) pick[3] = vi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)S;
    float sq = 0.f;
    for (int i = lane; i < S; i += 64) { const float d = v[i] - mean; sq += d * d; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    __syncthreads();
    if (lane == 0) {
        // torch.lerp(a, b, w): a + w*(b-a) for w < 0.5, else b - (b-a)*(1-w)
        const float w_lo = r_lo - (float)k0, w_hi = r_hi - (float)k2;
        const float lo = w_lo < 0.5f ? pick[0] + w_lo * (pick[1] - pick[0]) : pick[1] - (pick[1] - pick[0]) * (1.0f - w_lo);
        const float hi = w_hi < 0.5f ? pick[2] + w_hi * (pick[3] - pick[2]) : pick[3] - (pick[3] - pick[2]) * (1.0f - w_hi);
        piw[bc] = hi - lo;
        var[bc] = S > 1 ? sq / (float)(S - 1) : NAN;
    }
}

extern "C" int nd_sample_stats(const float* probs, float* piw, float* var, int S, int B, int C, float q_lo, float q_hi, void* stream) {
    if (!probs || !piw || !var) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (S < 1 || S > ND_STATS_MAXS || B < 1 || C < 1) return nd_set_err(ND_ERR_ARG, "need 1 <= S <= %d, B,C >= 1", ND_STATS_MAXS);
    if (!(q_lo >= 0.f && q_lo <= q_hi && q_hi <= 1.f)) return nd_set_err(ND_ERR_ARG, "need 0 <= q_lo <= q_hi <= 1");
    hipLaunchKernelGGL(k_sample_stats, dim3(B * C), dim3(64), 0, (hipStream_t)stream, probs, piw, var, S, B, C, q_lo, q_hi);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

// One thread per reported number, each a fixed-order loop over the N images (reproducible; N is a test-set size).
#define ND_REPORT_MAXBINS 64
__global__ __launch_bounds__(256) void k_report(const float* __restrict__ piw, const float* __restrict__ var, const float* __restrict__ pm,
                                                const long long* __restrict__ vote, const long long* __restrict__ target,
                                                float* __restrict__ out, int N, int C, float temperature, int n_bins) {
#pragma clang fp contract(off)
    const int tid = threadIdx.x;
    __shared__ float ece_part[ND_REPORT_MAXBINS];
    if (tid == 0) {                                         // accuracy
        int correct = 0;
        for (int i = 0; i < N; ++i) correct += vote[i] == target[i];
        out[0] = (float)correct / (float)N;
    }
    if (tid >= 1 && tid <= n_bins) {                        // calibration bin tid-1: (b[k], b[k+1]]
        const int k = tid - 1, steps = n_bins + 1;
        const float step = 1.0f / (float)n_bins;
        auto bound = [&](int i) { return i < steps / 2 ? step * (float)i : 1.0f - step * (float)(steps - 1 - i); };   // torch.linspace
        const float lo = bound(k), hi = bound(k + 1);
        float cnt = 0.f, csum = 0.f, asum = 0.f;
        for (int i = 0; i < N; ++i) {
            // convert_to_prob on the averaged probabilities (quirk kept), then confidence = max, prediction = argmax
            float mx = -INFINITY, lg[16];
            for (int c = 0; c < C; ++c) { const float d = pm[(size_t)i * C + c] - 1.0f; lg[c] = d * d * (-1.0f) / temperature; mx = fmaxf(mx, lg[c]); }
            float sum = 0.f;
            for (int c = 0; c < C; ++c) { lg[c] = expf(lg[c] - mx); sum += lg[c]; }
            float conf = -1.f; int pred = 0;
            for (int c = 0; c < C; ++c) { const float pc = lg[c] / sum; if (pc > conf) { conf = pc; pred = c; } }
            if (conf > lo && conf <= hi) { cnt += 1.f; csum += conf; asum += (pred == (int)target[i]) ? 1.f : 0.f; }
        }
        ece_part[k] = cnt > 0.f ? fabsf(asum / cnt - csum / cnt) * (cnt / (float)N) : 0.f;
    }
    const int q = tid - 128;                                // class statistics: q = kind * C + c
    if (q >= 0 && q < 4 * C) {
        const int kind = q / C, c = q % C;                  // 0 piw correct, 1 piw incorrect, 2 var correct, 3 var incorrect
        float sum = 0.f; int cnt = 0;
        for (int i = 0; i < N; ++i) {
            const int mv = (int)vote[i], gt = (int)target[i];
            if (mv != c) continue;
            const bool correct = mv == gt;
            if ((kind & 1) == (correct ? 0 : 1)) {
                sum += kind < 2 ? piw[(size_t)i * C + mv] : var[(size_t)i * C + c];
                ++cnt;
            }
        }
        out[2 + q] = kind < 2 ? sum / (float)cnt : (cnt > 0 ? sum / (float)cnt : 0.f);   // empty: NaN for PIW, 0 for variance
    }
    __syncthreads();
    if (tid == 0) {
        float e = 0.f;
        for (int k = 0; k < n_bins; ++k) e += ece_part[k];
        out[1] = e;
    }
}

extern "C" int nd_report(const float* piw, const float* var, const float* pm, const int64_t* vote, const int64_t* target, float* out,
                         int N, int C, float temperature, int n_bins, void* stream) {
    if (!piw || !var || !pm || !vote || !target || !out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (N < 1 || C < 1 || C > 16 || n_bins < 1 || n_bins > ND_REPORT_MAXBINS) return nd_set_err(ND_ERR_ARG, "need N >= 1, 1 <= C <= 16, 1 <= n_bins <= %d", ND_REPORT_MAXBINS);
    if (!(temperature > 0.f)) return nd_set_err(ND_ERR_ARG, "temperature must be > 0");
    hipLaunchKernelGGL(k_report, dim3(1), dim3(256), 0, (hipStream_t)stream, piw, var, pm, (const long long*)vote, (const long long*)target, out,
                       N, C, temperature, n_bins);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}
