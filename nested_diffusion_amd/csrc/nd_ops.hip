// nd_ops.hip -- standalone operators: skinny Linear (mapping MLP), row softmax, ensemble aggregation.
// gfx950 only.
#include "nd_common.hpp"
#include "../../include/nested_diffusion.h"

int nd_set_err(int code, const char* fmt, ...);
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return nd_set_err(ND_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ---- packing ---------------------------------------------------------------------------------
extern "C" size_t nd_packed_bytes(int R, int K) {
    if (R < 1 || K < 16 || (K % 16)) return 0;
    return nd_packed_floats(R, K) * sizeof(float);
}

extern "C" int nd_pack_rows(const float* src, float* dst, int R, int K, void* stream) {
    if (!src || !dst) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (R < 1 || K < 16 || (K % 16)) return nd_set_err(ND_ERR_ARG, "need R >= 1 and K a positive multiple of 16 (K=%d)", K);
    const size_t n4 = nd_packed_floats(R, K) / 4, want = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)(want > 8192 ? 8192 : want)), dim3(256), 0, (hipStream_t)stream, src, dst, R, K);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}

// ---- nd_linear: mapping/models/mlp.py:25-28 ---------------------------------------------------
extern "C" size_t nd_linear_workspace_bytes(int M, int K, int N) {
    if (M < 1 || K < 16 || (K % 16) || N < 1) return 0;
    size_t fl = nd_packed_floats(M, K) + 64;
    if (nd_use_splitk(K)) fl += nd_splitk_part_floats(M, K, N);
    return fl * sizeof(float) + 256;
}

extern "C" int nd_linear(const float* x, const float* wpk, const float* scale, const float* shift, float* out, int M, int K, int N,
                         int act, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !wpk || !out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (M < 1 || N < 1 || K < 16 || (K % 16)) return nd_set_err(ND_ERR_ARG, "need M,N >= 1 and K a positive multiple of 16 (K=%d)", K);
    if (act < 0 || act > 3) return nd_set_err(ND_ERR_ARG, "unknown activation %d", act);
    const size_t need = nd_linear_workspace_bytes(M, K, N);
    if (!ws || ws_bytes < need) return nd_set_err(ND_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    float* xpk = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* part = xpk + nd_packed_floats(M, K) + 64;
    int rc = nd_pack_rows(x, xpk, M, K, stream);
    if (rc != ND_OK) return rc;
    if (nd_use_splitk(K)) {
        const SkinnyLaunch L = nd_skinny_launch<2>(K, N, M, 1);
        SkinnyDesc sd{xpk, wpk, nullptr, nullptr, nullptr, nullptr, part, K, N, 0, ND_ACT_NONE, 0};
        HIP_CHECK(nd_launch_skinny(L, sd, nullptr, 1, M, 0, st));
        SplitKEpiDesc se{part, scale, shift, out, N, L.S, act, 0};
        const size_t q = (size_t)(((M + 15) / 16) * 16) * (((N + 15) / 16) * 16) / 4;
        hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, se, (const SplitKEpiDesc*)nullptr, M, L.S);
    } else {
        SkinnyDesc d{xpk, wpk, scale, shift, out, nullptr, nullptr, K, N, 0, act, 0};
        HIP_CHECK(nd_launch_skinny(nd_skinny_launch<0>(K, N, M, 1), d, nullptr, 1, M, 0, st));
    }
    HIP_CHECK(hipGetLastError());
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
