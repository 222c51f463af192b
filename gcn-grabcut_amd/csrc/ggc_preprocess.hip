// ggc_preprocess.hip — G0: colour preparation of GraphBuilder.__init__
// (reference graph_builder.py:142-154): BGR -> Lab (f64 arithmetic, f32 store),
// BGR -> HSV, BGR -> GRAY (OpenCV 4.x 15-bit fixed point), Sobel 3x3 magnitude.
//
// One thread per pixel, one launch per batch.  HBM traffic per pixel: 3 B read
// (+ cached 3x3 neighbourhood for the Sobel) and 32 B written; the kernel is
// bandwidth-trivial and bound by the f64 Newton iterations of the Lab transfer
// function (4.2 MB per 400x300 image).
#include "ggc_internal.h"
#include "ggc_math.h"

namespace ggc {

__device__ __forceinline__ double lab_f(double t) {
    return t > 0.008856 ? det_cbrt(t) : 7.787 * t + 16.0 / 116.0;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

__device__ __forceinline__ float gray_at(const uint8_t* __restrict__ img, int W, int y, int x) {
    const uint8_t* p = img + ((size_t)y * W + x) * 3;
    return (float)((p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15);
}

__global__ void __launch_bounds__(256) k_preprocess(int H, int W, const uint8_t* __restrict__ bgr,
                                                    const double* __restrict__ lut,
                                                    float* __restrict__ lab, float* __restrict__ hsv,
                                                    float* __restrict__ gray, float* __restrict__ grad) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t P = (size_t)H * W;
    const uint8_t* img = bgr + (size_t)blockIdx.z * P * 3;
    const size_t p = (size_t)blockIdx.z * P + (size_t)y * W + x;
    const uint8_t* px = img + ((size_t)y * W + x) * 3;
    const int b8 = px[0], g8 = px[1], r8 = px[2];
    if (lab) {
        const double R = lut[r8], G = lut[g8], B = lut[b8];
        const double X = R * 0.412453 + G * 0.357580 + B * 0.180423;
        const double Y = R * 0.212671 + G * 0.715160 + B * 0.072169;
        const double Z = R * 0.019334 + G * 0.119193 + B * 0.950227;
        const double fx = lab_f(X / 0.95047), fy = lab_f(Y / 1.0), fz = lab_f(Z / 1.08883);
        lab[3 * p + 0] = (float)(116.0 * fy - 16.0);
        lab[3 * p + 1] = (float)(500.0 * (fx - fy));
        lab[3 * p + 2] = (float)(200.0 * (fy - fz));
    }
    if (hsv) {
        const double r = r8 * (1.0 / 255.0), g = g8 * (1.0 / 255.0), b = b8 * (1.0 / 255.0);
        const double v = fmax(fmax(r, g), b);
        const double mn = fmin(fmin(r, g), b);
        const double delta = v - mn;
        double s = 0.0, h = 0.0;
        if (delta != 0.0) {
            s = delta / v;
            if (r == v) h = (g - b) / delta;            // later matches override earlier ones
            if (g == v) h = 2.0 + (b - r) / delta;
            if (b == v) h = 4.0 + (r - g) / delta;
            h = h / 6.0;
            if (h < 0.0) h = h + 1.0;
        }
        hsv[3 * p + 0] = (float)h;
        hsv[3 * p + 1] = (float)s;
        hsv[3 * p + 2] = (float)v;
    }
    if (gray) gray[p] = (float)((b8 * 3735 + g8 * 19235 + r8 * 9798 + (1 << 14)) >> 15);
    if (grad) {
        const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
        const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
        const float a = gray_at(img, W, ym, xm), b = gray_at(img, W, ym, x), c = gray_at(img, W, ym, xp);
        const float d = gray_at(img, W, y, xm), f = gray_at(img, W, y, xp);
        const float g = gray_at(img, W, yp, xm), h = gray_at(img, W, yp, x), i = gray_at(img, W, yp, xp);
        const float gx = (c + 2.0f * f + i) - (a + 2.0f * d + g);
        const float gy = (g + 2.0f * h + i) - (a + 2.0f * b + c);
        grad[p] = sqrtf(gx * gx + gy * gy);
    }
}

} // namespace ggc

namespace ggc {
// C7: 8-bit HSV / Lab for GrabCutConfig.color_space (reference grabcut.py:73-79): integer / float64 sequences that the
// CPU checker restates operation for operation; OpenCV parity unpinned.  `lut` = sRGB -> linear table (mode 1), `hdiv` / `sdiv`
// = OpenCV's fixed-point reciprocal tables (mode 0).
__global__ void __launch_bounds__(256) k_convert_color8(size_t n, const uint8_t* __restrict__ bgr, int mode,
                                                        const double* __restrict__ lut, const int32_t* __restrict__ sdiv,
                                                        const int32_t* __restrict__ hdiv, uint8_t* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int b = bgr[3 * p + 0], g = bgr[3 * p + 1], r = bgr[3 * p + 2];
    if (mode == 0) {
        const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
        const int s = (diff * sdiv[v] + (1 << 11)) >> 12;
        int h = (v == r) ? (g - b) : (v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff);
        h = (h * hdiv[diff] + (1 << 11)) >> 12;
        if (h < 0) h += 180;
        out[3 * p + 0] = (uint8_t)h; out[3 * p + 1] = (uint8_t)s; out[3 * p + 2] = (uint8_t)v;
        return;
    }
    const double R = lut[r], G = lut[g], B = lut[b];
    const double X = R * 0.412453 + G * 0.357580 + B * 0.180423;
    const double Y = R * 0.212671 + G * 0.715160 + B * 0.072169;
    const double Z = R * 0.019334 + G * 0.119193 + B * 0.950227;
    const double tx = X / 0.95047, ty = Y / 1.0, tz = Z / 1.08883;
    const double fx = tx > 0.008856 ? det_cbrt(tx) : 7.787 * tx + 16.0 / 116.0;
    const double fy = ty > 0.008856 ? det_cbrt(ty) : 7.787 * ty + 16.0 / 116.0;
    const double fz = tz > 0.008856 ? det_cbrt(tz) : 7.787 * tz + 16.0 / 116.0;
    const double v[3] = {(116.0 * fy - 16.0) * 255.0 / 100.0, 500.0 * (fx - fy) + 128.0, 200.0 * (fy - fz) + 128.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double q = floor(v[c] + 0.5);
        q = q < 0.0 ? 0.0 : (q > 255.0 ? 255.0 : q);
        out[3 * p + c] = (uint8_t)q;
    }
}
} // namespace ggc

using namespace ggc;

extern "C" int ggc_preprocess(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const uint8_t* bgr,
                              float* lab, float* hsv, float* gray, float* grad) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, bgr != nullptr, GGC_E_INVALID_ARG, "bgr is NULL");
    GGC_REQUIRE(ctx, B <= 65535, GGC_E_SHAPE, "batch %d exceeds the grid z limit", B);
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // sRGB -> linear table for the 256 byte values (skimage: u * (1/255.0), colorconv.py:657-661)
    double* lut = scratch_t<double>(ctx, S_PRE_A, 256);
    if (!lut) return GGC_E_OOM;
    static thread_local double host_lut[256];
    static thread_local bool lut_ready = false;
    if (!lut_ready) {
        for (int u = 0; u < 256; ++u) {
            const double v = (double)u * (1.0 / 255.0);
            host_lut[u] = v > 0.04045 ? det_pow24((v + 0.055) / 1.055) : v / 12.92;
        }
        lut_ready = true;
    }
    GGC_HIP(ctx, hipMemcpyAsync(lut, host_lut, sizeof(host_lut), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_preprocess, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(256), 0, st, H, W, bgr, lut, lab, hsv,
                       gray, grad);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

extern "C" int ggc_convert_color8(ggc_ctx* ctx, ggc_stream stream, int64_t n_pixels, const uint8_t* bgr, int mode, uint8_t* out) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, n_pixels >= 1 && bgr && out, GGC_E_INVALID_ARG, "bad arguments");
    GGC_REQUIRE(ctx, mode == 0 || mode == 1, GGC_E_INVALID_ARG, "mode %d: 0 = HSV, 1 = Lab", mode);
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    double* lut = scratch_t<double>(ctx, S_PRE_A, 256);
    int32_t* tabs = scratch_t<int32_t>(ctx, S_PRE_B, 512);
    if (!lut || !tabs) return GGC_E_OOM;
    double host_lut[256];
    int32_t host_tabs[512];
    for (int u = 0; u < 256; ++u) {
        const double v = (double)u * (1.0 / 255.0);
        host_lut[u] = v > 0.04045 ? det_pow24((v + 0.055) / 1.055) : v / 12.92;
        host_tabs[u] = u ? (int32_t)rint((double)(255 << 12) / (double)u) : 0;                 // sdiv_table
        host_tabs[256 + u] = u ? (int32_t)rint((double)(180 << 12) / (6.0 * (double)u)) : 0;   // hdiv_table (H in [0,180))
    }
    GGC_HIP(ctx, hipMemcpyAsync(lut, host_lut, sizeof(host_lut), hipMemcpyHostToDevice, st));
    GGC_HIP(ctx, hipMemcpyAsync(tabs, host_tabs, sizeof(host_tabs), hipMemcpyHostToDevice, st));
    GGC_HIP(ctx, hipStreamSynchronize(st));                                                     // the tables live on this stack frame
    hipLaunchKernelGGL(k_convert_color8, dim3(cdiv((size_t)n_pixels, 256)), dim3(256), 0, st, (size_t)n_pixels, bgr, mode, lut, tabs,
                       tabs + 256, out);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}
