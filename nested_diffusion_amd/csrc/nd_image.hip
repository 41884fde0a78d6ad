// nd_image.hip -- input-side perturbations of the robustness protocol (diffusion/utils.py:272-414, applied at
// classification_train_separately.py:726-737).  Elementwise / resampling kernels on [B, C, H, W] fp32 images.
// gfx950 only.
#include <hip/hip_runtime.h>
#include "../../include/nested_diffusion.h"

int nd_set_err(int code, const char* fmt, ...);
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return nd_set_err(ND_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

static inline dim3 grid1(size_t n, int per = 256) {
    const size_t b = (n + per - 1) / per;
    return dim3((unsigned)(b > 65535 * 16 ? 65535 * 16 : (b ? b : 1)));
}

// out = x + z * std           (add_noise, utils.py:272-279; z = the torch.randn_like draw, supplied)
__global__ void k_add_noise(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ out, size_t n, float std_) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = x[i] + z[i] * std_;
}
// out = clamp(x + k, 0, 1)    (adjust_brightness, utils.py:390-399)
__global__ void k_brightness(const float* __restrict__ x, float* __restrict__ out, size_t n, float k) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf(x[i] + k, 0.f), 1.f);
}
// per-image mean over C*H*W (adjust_contrast, utils.py:405-406): one workgroup per image, fixed reduction tree
__global__ __launch_bounds__(1024) void k_image_mean(const float* __restrict__ x, float* __restrict__ mean, size_t per) {
    const float* p = x + (size_t)blockIdx.x * per;
    float s = 0.f;
    for (size_t i = threadIdx.x; i < per; i += 1024) s += p[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ float red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        mean[blockIdx.x] = t / (float)per;
    }
}
// out = clamp(mean + (x - mean) * k, 0, 1)   (utils.py:408-412)
__global__ void k_contrast(const float* __restrict__ x, const float* __restrict__ mean, float* __restrict__ out, size_t per, size_t n, float k) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float m = mean[i / per];
        out[i] = fminf(fmaxf(m + (x[i] - m) * k, 0.f), 1.f);
    }
}

// Bilinear resize, torch.nn.functional.interpolate(mode='bilinear', align_corners=False, no antialias) semantics:
// src = max(scale * (dst + 0.5) - 0.5, 0), scale = in / out; neighbours clamped at the border.
// Optional per-image crop window (top, left, size x size) read from `crop` (random_crop_and_resize, utils.py:282-312).
__global__ void k_resize_bilinear(const float* __restrict__ x, float* __restrict__ out, int NC, int C, int Hi, int Wi, int Ho, int Wo,
                                  const int* __restrict__ crop, int crop_size) {
#pragma clang fp contract(off)
    const size_t total = (size_t)NC * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), nc = (int)(i / ((size_t)Wo * Ho));
        int top = 0, left = 0, hi = Hi, wi = Wi;
        if (crop) { const int b = nc / C; top = crop[2 * b]; left = crop[2 * b + 1]; hi = crop_size; wi = crop_size; }
        const float sh = (float)hi / (float)Ho, sw = (float)wi / (float)Wo;
        const float fy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)fy, hi - 1), x0 = min((int)fx, wi - 1);
        const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* p = x + (size_t)nc * Hi * Wi;
        const float v00 = p[(size_t)(top + y0) * Wi + left + x0], v01 = p[(size_t)(top + y0) * Wi + left + x1];
        const float v10 = p[(size_t)(top + y1) * Wi + left + x0], v11 = p[(size_t)(top + y1) * Wi + left + x1];
        out[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}

// zero `n_rects` squares of side `side` per image (random_cover_new, utils.py:315-349); rects[b][r] = (top, left)
__global__ void k_cover(float* __restrict__ x, int B, int C, int H, int W, const int* __restrict__ rects, int n_rects, int side) {
    const size_t per = (size_t)n_rects * C * side * side, total = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int dx = (int)(i % side), dy = (int)((i / side) % side), c = (int)((i / ((size_t)side * side)) % C);
        const int r = (int)((i / ((size_t)side * side * C)) % n_rects), b = (int)(i / per);
        const int top = rects[((size_t)b * n_rects + r) * 2], left = rects[((size_t)b * n_rects + r) * 2 + 1];
        x[(((size_t)b * C + c) * H + top + dy) * W + left + dx] = 0.f;
    }
}

extern "C" int nd_img_add_noise(const float* x, const float* z, float* out, size_t n, float std_, void* stream) {
    if (!x || !z || !out || n == 0) return nd_set_err(ND_ERR_ARG, "bad add_noise arguments");
    hipLaunchKernelGGL(k_add_noise, grid1(n), dim3(256), 0, (hipStream_t)stream, x, z, out, n, std_);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}
extern "C" int nd_img_brightness(const float* x, float* out, size_t n, float k, void* stream) {
    if (!x || !out || n == 0) return nd_set_err(ND_ERR_ARG, "bad brightness arguments");
    hipLaunchKernelGGL(k_brightness, grid1(n), dim3(256), 0, (hipStream_t)stream, x, out, n, k);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}
extern "C" int nd_img_contrast(const float* x, float* out, float* mean_ws, int B, size_t per_image, float k, void* stream) {
    if (!x || !out || !mean_ws || B < 1 || per_image == 0) return nd_set_err(ND_ERR_ARG, "bad contrast arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_image_mean, dim3(B), dim3(1024), 0, st, x, mean_ws, per_image);
    const size_t n = (size_t)B * per_image;
    hipLaunchKernelGGL(k_contrast, grid1(n), dim3(256), 0, st, x, (const float*)mean_ws, out, per_image, n, k);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}
extern "C" int nd_img_resize_bilinear(const float* x, float* out, int B, int C, int Hi, int Wi, int Ho, int Wo, const int32_t* crop,
                                      int crop_size, void* stream) {
    if (!x || !out || B < 1 || C < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return nd_set_err(ND_ERR_ARG, "bad resize arguments");
    if (crop && (crop_size < 1 || crop_size > Hi || crop_size > Wi)) return nd_set_err(ND_ERR_ARG, "crop_size out of range");
    const size_t total = (size_t)B * C * Ho * Wo;
    hipLaunchKernelGGL(k_resize_bilinear, grid1(total), dim3(256), 0, (hipStream_t)stream, x, out, B * C, C, Hi, Wi, Ho, Wo, (const int*)crop, crop_size);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}
extern "C" int nd_img_cover(float* x, int B, int C, int H, int W, const int32_t* rects, int n_rects, int side, void* stream) {
    if (!x || !rects || B < 1 || C < 1 || n_rects < 1 || side < 0 || side > H || side > W) return nd_set_err(ND_ERR_ARG, "bad cover arguments");
    if (side == 0) return ND_OK;
    const size_t total = (size_t)B * n_rects * C * side * side;
    hipLaunchKernelGGL(k_cover, grid1(total), dim3(256), 0, (hipStream_t)stream, x, B, C, H, W, (const int*)rects, n_rects, side);
    HIP_CHECK(hipGetLastError());
    return ND_OK;
}
