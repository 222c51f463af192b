/*
 * oracle/color.c — colour preparation of GraphBuilder.__init__.
 * TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Follows reference src/gcn_grabcut/graph_builder.py:142-154:
 *   rgb  = cv2.cvtColor(bgr, BGR2RGB)
 *   lab  = skimage.color.rgb2lab(rgb).astype(float32)      (f64 arithmetic)
 *   hsv  = skimage.color.rgb2hsv(rgb).astype(float32)      (f64 arithmetic)
 *   gray = cv2.cvtColor(bgr, BGR2GRAY).astype(float32)     (8-bit fixed point)
 *   grad = sqrt(Sobel_x(gray)^2 + Sobel_y(gray)^2)         (CV_32F, ksize 3, REFLECT_101)
 * with the third-party semantics of SURVEY.md Appendix A (skimage 0.18.3
 * colorconv.py:190-270, 622-663, 906-970; img_as_float(uint8) = u * (1/255.0);
 * OpenCV 4.x 15-bit grey coefficients).  pow(.,2.4) and cbrt are the
 * deterministic restatements of mathfn.c; the 3x3 matrix product is evaluated
 * left to right without FMA.
 */
#include "ggc_oracle.h"
#include <math.h>
#include <stddef.h>

static double srgb_to_linear(int u) {
    double v = (double)u * (1.0 / 255.0);
    return v > 0.04045 ? ggo_pow24((v + 0.055) / 1.055) : v / 12.92;
}

static double lab_f(double t) {
    return t > 0.008856 ? ggo_cbrt(t) : 7.787 * t + 16.0 / 116.0;
}

static int reflect101(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

void ggo_preprocess(int H, int W, const uint8_t* bgr,
                    float* lab, float* hsv, float* gray, float* grad) {
    double lut[256];
    for (int u = 0; u < 256; ++u) lut[u] = srgb_to_linear(u);
    const size_t P = (size_t)H * W;
    for (size_t p = 0; p < P; ++p) {
        const int b8 = bgr[3 * p + 0], g8 = bgr[3 * p + 1], r8 = bgr[3 * p + 2];
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
            double v = r > g ? r : g; if (b > v) v = b;
            double mn = r < g ? r : g; if (b < mn) mn = b;
            const double delta = v - mn;
            double s = 0.0, h = 0.0;
            if (delta != 0.0) {
                s = delta / v;
                /* later assignments override earlier ones (colorconv.py:243-252) */
                if (r == v) h = (g - b) / delta;
                if (g == v) h = 2.0 + (b - r) / delta;
                if (b == v) h = 4.0 + (r - g) / delta;
                h = h / 6.0;
                if (h < 0.0) h = h + 1.0;      /* (h / 6) % 1 for h/6 in [-1/6, 5/6] */
            }
            hsv[3 * p + 0] = (float)h;
            hsv[3 * p + 1] = (float)s;
            hsv[3 * p + 2] = (float)v;
        }
        if (gray) gray[p] = (float)((b8 * 3735 + g8 * 19235 + r8 * 9798 + (1 << 14)) >> 15);
    }
    if (grad && gray) {
        for (int y = 0; y < H; ++y) {
            const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
            for (int x = 0; x < W; ++x) {
                const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
                const float a = gray[(size_t)ym * W + xm], b = gray[(size_t)ym * W + x], c = gray[(size_t)ym * W + xp];
                const float d = gray[(size_t)y * W + xm], f = gray[(size_t)y * W + xp];
                const float g = gray[(size_t)yp * W + xm], h = gray[(size_t)yp * W + x], i = gray[(size_t)yp * W + xp];
                const float gx = (c + 2.0f * f + i) - (a + 2.0f * d + g);
                const float gy = (g + 2.0f * h + i) - (a + 2.0f * b + c);
                grad[(size_t)y * W + x] = sqrtf(gx * gx + gy * gy);
            }
        }
    }
}

/* 8-bit colour spaces for GrabCutConfig.color_space (reference grabcut.py:73-79 calls cv2.cvtColor(BGR2HSV / BGR2Lab) on
 * uint8 images).  OpenCV is absent: PARITY UNPINNED.
 *   mode 0, HSV: OpenCV's integer RGB2HSV_b as recalled from color_hsv.simd.hpp — H in [0,180), fixed point with
 *     sdiv_table[v] = round((255 << 12) / v), hdiv_table[d] = round((180 << 12) / (6 d)).
 *   mode 1, Lab: the documented 8-bit convention L <- L*255/100, a <- a+128, b <- b+128 applied to the float64 CIELAB of
 *     ggo_preprocess (D65, same matrix), rounded half up and clamped; OpenCV's own fixed-point tables may differ by one
 *     level. */
void ggo_convert_color8(size_t n, const uint8_t* bgr, int mode, uint8_t* out) {
    if (mode == 0) {
        for (size_t p = 0; p < n; ++p) {
            const int b = bgr[3 * p + 0], g = bgr[3 * p + 1], r = bgr[3 * p + 2];
            int v = b > g ? b : g; if (r > v) v = r;
            int vmin = b < g ? b : g; if (r < vmin) vmin = r;
            const int diff = v - vmin;
            const int sdiv = v ? (int)rint((double)(255 << 12) / (double)v) : 0;
            const int hdiv = diff ? (int)rint((double)(180 << 12) / (6.0 * (double)diff)) : 0;
            const int s = (diff * sdiv + (1 << 11)) >> 12;
            int h = (v == r) ? (g - b) : (v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff);
            h = (h * hdiv + (1 << 11)) >> 12;
            if (h < 0) h += 180;
            out[3 * p + 0] = (uint8_t)h; out[3 * p + 1] = (uint8_t)s; out[3 * p + 2] = (uint8_t)v;
        }
        return;
    }
    double lut[256];
    for (int u = 0; u < 256; ++u) lut[u] = srgb_to_linear(u);
    for (size_t p = 0; p < n; ++p) {
        const double B = lut[bgr[3 * p + 0]], G = lut[bgr[3 * p + 1]], R = lut[bgr[3 * p + 2]];
        const double X = R * 0.412453 + G * 0.357580 + B * 0.180423;
        const double Y = R * 0.212671 + G * 0.715160 + B * 0.072169;
        const double Z = R * 0.019334 + G * 0.119193 + B * 0.950227;
        const double fx = lab_f(X / 0.95047), fy = lab_f(Y / 1.0), fz = lab_f(Z / 1.08883);
        const double v[3] = {(116.0 * fy - 16.0) * 255.0 / 100.0, 500.0 * (fx - fy) + 128.0, 200.0 * (fy - fz) + 128.0};
        for (int c = 0; c < 3; ++c) {
            double q = floor(v[c] + 0.5);
            if (q < 0.0) q = 0.0;
            if (q > 255.0) q = 255.0;
            out[3 * p + c] = (uint8_t)q;
        }
    }
}
