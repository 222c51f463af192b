/*
 * oracle/slic.c — skimage.segmentation.slic as the reference calls it.
 * TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Reference call site: src/gcn_grabcut/graph_builder.py:177-188
 *     slic(lab_f32, n_segments, compactness=10, sigma=1, start_label=0, channel_axis=-1)
 * Third-party semantics restated from SURVEY.md Appendix A.1 and the
 * scikit-image 0.18.3 sources in this container
 * (segmentation/slic_superpixels.py:107-330, util/_regular_grid.py:61-83,
 * color/colorconv.py:622-663, 906-970) plus scipy.ndimage.gaussian_filter.
 *
 * Pinned bit-exact against the compiled 0.18.3 kernels (_slic_cython,
 * _enforce_label_connectivity_cython) and scipy's gaussian_filter by
 * tests/test_slic_oracle.py with the fixtures in tests/golden/.
 * The ">= 0.19" global min-max rescale (rescale_input) is recalled from the
 * upstream source and cannot be verified here: PARITY UNPINNED for that step.
 *
 * All image arithmetic is float32 with one rounding per operation unless a
 * comment says f64.
 */
#include "ggc_oracle.h"
#include "../include/ggc_fmath.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- step 1-2: rescale to [0,1] (skimage >= 0.19) and second rgb2lab in f32 */
void ggo_slic_rescale_lab(int H, int W, const float* image, int rescale_input, float* out) {
    const size_t n = (size_t)H * W * 3;
    float mn = image[0], mx = image[0];
    for (size_t i = 1; i < n; ++i) { if (image[i] < mn) mn = image[i]; if (image[i] > mx) mx = image[i]; }
    const float range = mx - mn;
    const float thr_rgb = (float)0.04045, thr_lab = (float)0.008856;
    const float m[9] = {(float)0.412453, (float)0.357580, (float)0.180423,
                        (float)0.212671, (float)0.715160, (float)0.072169,
                        (float)0.019334, (float)0.119193, (float)0.950227};
    const float wx = (float)0.95047, wy = 1.0f, wz = (float)1.08883;
    const float k7 = (float)7.787, k16 = (float)(16.0 / 116.0);
    for (size_t p = 0; p < (size_t)H * W; ++p) {
        float lin[3];
        for (int c = 0; c < 3; ++c) {
            float v = image[3 * p + c];
            if (rescale_input) { v = v - mn; if (mx != mn) v = v / range; }
            lin[c] = v > thr_rgb ? (float)ggo_pow24((double)((v + (float)0.055) / (float)1.055))
                                 : v / (float)12.92;
        }
        float xyz[3];
        for (int r = 0; r < 3; ++r) xyz[r] = (lin[0] * m[3 * r] + lin[1] * m[3 * r + 1]) + lin[2] * m[3 * r + 2];
        float t[3] = {xyz[0] / wx, xyz[1] / wy, xyz[2] / wz};
        float f[3];
        for (int c = 0; c < 3; ++c)
            f[c] = t[c] > thr_lab ? (float)ggo_cbrt((double)t[c]) : k7 * t[c] + k16;
        out[3 * p + 0] = 116.0f * f[1] - 16.0f;
        out[3 * p + 1] = 500.0f * (f[0] - f[1]);
        out[3 * p + 2] = 200.0f * (f[1] - f[2]);
    }
}

/* ---- step 4: scipy.ndimage.gaussian_filter(image, [sigma, sigma, 0]) on (H,W,C) f32 */
static double np_pairwise_sum(const double* a, int n) {
    /* numpy's pairwise summation for n <= 128 (PW_BLOCKSIZE) */
    if (n < 8) { double s = 0.0; for (int i = 0; i < n; ++i) s += a[i]; return s; }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

static int reflect_sym(int i, int n) { /* scipy mode='reflect': (d c b a | a b c d | d c b a) */
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period; if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

int ggo_gaussian_weights(double sigma, double* w /*[2r+1]*/, int cap) {
    const int r = (int)(4.0 * sigma + 0.5);
    if (2 * r + 1 > cap) return -1;
    ggc_gaussian_taps(sigma, r, w);                     /* numpy's exp values at sigma = 1 (include/ggc_fmath.h) */
    const double sum = np_pairwise_sum(w, 2 * r + 1);
    for (int i = 0; i < 2 * r + 1; ++i) w[i] = w[i] / sum;
    return r;
}

static void gauss_axis(int H, int W, int C, const float* in, float* out, const double* w, int r, int axis) {
    const int len = axis == 0 ? H : W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < C; ++c) {
                const int l = axis == 0 ? y : x;
#define AT(idx) ((double)(axis == 0 ? in[((size_t)reflect_sym((idx), len) * W + x) * C + c] \
                                    : in[((size_t)y * W + reflect_sym((idx), len)) * C + c]))
                double tmp = AT(l) * w[r];
                for (int jj = -r; jj < 0; ++jj) tmp += (AT(l + jj) + AT(l - jj)) * w[jj + r];
#undef AT
                out[((size_t)y * W + x) * C + c] = (float)tmp;
            }
}

void ggo_gaussian_f32(int H, int W, int C, const float* in, double sigma, float* out) {
    double w[129];
    const int r = ggo_gaussian_weights(sigma, w, 129);
    float* tmp = (float*)malloc((size_t)H * W * C * sizeof(float));
    gauss_axis(H, W, C, in, tmp, w, r, 0);   /* axis y first */
    gauss_axis(H, W, C, tmp, out, w, r, 1);  /* then x */
    free(tmp);
}

/* ---- step 3: regular_grid((1,H,W), n) (util/_regular_grid.py:61-83) */
int ggo_slic_grid(int H, int W, int n_segments, int* step_y, int* step_x,
                  int* start_y, int* start_x, int* ny, int* nx) {
    /* sorted dims of (1,H,W) and where H, W land */
    double dims[3] = {1.0, (double)(H <= W ? H : W), (double)(H <= W ? W : H)};
    double space = 1.0 * H * W;
    if (space <= (double)n_segments) { /* every pixel is a seed */
        *step_y = *step_x = 1; *start_y = *start_x = 0; *ny = H; *nx = W;
        return H * W;
    }
    double st[3];
    for (int i = 0; i < 3; ++i) st[i] = pow(space / n_segments, 1.0 / 3.0);
    if (dims[0] < st[0] || dims[1] < st[1] || dims[2] < st[2]) {
        for (int dim = 0; dim < 3; ++dim) {
            st[dim] = dims[dim];
            double sp = 1.0;
            for (int j = dim + 1; j < 3; ++j) sp *= dims[j];
            for (int j = dim + 1; j < 3; ++j) st[j] = pow(sp / n_segments, 1.0 / (3 - dim - 1));
            if (dims[0] >= st[0] && dims[1] >= st[1] && dims[2] >= st[2]) break;
        }
    }
    int starts[3], steps[3];
    for (int i = 0; i < 3; ++i) { starts[i] = (int)floor(st[i] / 2.0); steps[i] = (int)rint(st[i]); }
    const int iy = (H <= W) ? 1 : 2, ix = (H <= W) ? 2 : 1;
    *step_y = steps[iy]; *start_y = starts[iy];
    *step_x = steps[ix]; *start_x = starts[ix];
    *ny = (H - *start_y + *step_y - 1) / *step_y;
    *nx = (W - *start_x + *step_x - 1) / *step_x;
    if (*ny < 0) *ny = 0;
    if (*nx < 0) *nx = 0;
    return *ny * *nx;
}

/* ---- step 6: _slic_cython (k-means), depth 1, spacing 1, no mask, not SLICO */
void ggo_slic_kmeans(int H, int W, const float* image, int K, float* centers,
                     float step, int max_iter, int32_t* labels) {
    int wy, wx, d0, d1, d2, d3;
    ggo_slic_grid(H, W, K, &wy, &wx, &d0, &d1, &d2, &d3); /* window steps from the ACTUAL seed count */
    const float sw = (float)(1.0 / (double)(step * step));
    const size_t P = (size_t)H * W;
    float* dist = (float*)malloc(P * sizeof(float));
    int64_t* cnt = (int64_t*)malloc((size_t)K * sizeof(int64_t));
    for (size_t p = 0; p < P; ++p) labels[p] = 0;
    for (int it = 0; it < max_iter; ++it) {
        int change = 0;
        for (size_t p = 0; p < P; ++p) dist[p] = INFINITY;
        for (int k = 0; k < K; ++k) {
            const float cy = centers[5 * k + 0], cx = centers[5 * k + 1];
            if (cy != cy || cx != cx) continue; /* dead seed (0/0 centroid) never matches again */
            float fy0 = cy - (float)(2 * wy); if (!(fy0 > 0.0f)) fy0 = 0.0f;
            float fy1 = cy + (float)(2 * wy) + 1.0f; if (!(fy1 < (float)H)) fy1 = (float)H;
            float fx0 = cx - (float)(2 * wx); if (!(fx0 > 0.0f)) fx0 = 0.0f;
            float fx1 = cx + (float)(2 * wx) + 1.0f; if (!(fx1 < (float)W)) fx1 = (float)W;
            const int y0 = (int)fy0, y1 = (int)fy1, x0 = (int)fx0, x1 = (int)fx1;
            const float c0 = centers[5 * k + 2], c1 = centers[5 * k + 3], c2 = centers[5 * k + 4];
            for (int y = y0; y < y1; ++y) {
                const float ty = cy - (float)y;
                const float dy = ty * ty;
                for (int x = x0; x < x1; ++x) {
                    const float tx = cx - (float)x;
                    float d = (dy + tx * tx) * sw;
                    const float* px = image + ((size_t)y * W + x) * 3;
                    float dc = 0.0f, t;
                    t = px[0] - c0; dc += t * t;
                    t = px[1] - c1; dc += t * t;
                    t = px[2] - c2; dc += t * t;
                    d += dc;
                    const size_t p = (size_t)y * W + x;
                    if (dist[p] > d) { labels[p] = k; dist[p] = d; change = 1; }
                }
            }
        }
        if (!change) break;
        memset(cnt, 0, (size_t)K * sizeof(int64_t));
        for (int k = 0; k < 5 * K; ++k) centers[k] = 0.0f;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t p = (size_t)y * W + x;
                const int k = labels[p];
                cnt[k] += 1;
                centers[5 * k + 0] += (float)y;
                centers[5 * k + 1] += (float)x;
                centers[5 * k + 2] += image[3 * p + 0];
                centers[5 * k + 3] += image[3 * p + 1];
                centers[5 * k + 4] += image[3 * p + 2];
            }
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 5; ++c) centers[5 * k + c] = centers[5 * k + c] / (float)cnt[k];
    }
    free(dist); free(cnt);
}

/* ---- step 7: _enforce_label_connectivity_cython, depth 1, start_label 0 */
int ggo_slic_connectivity(int H, int W, const int32_t* labels, int min_size, int max_size,
                          int32_t* out) {
    const size_t P = (size_t)H * W;
    const int ddx[4] = {1, -1, 0, 0};
    const int ddy[4] = {0, 0, 1, -1};
    int32_t* q = (int32_t*)malloc((size_t)(max_size > 0 ? max_size : 1) * 2 * sizeof(int32_t));
    for (size_t p = 0; p < P; ++p) out[p] = -1;
    int cur = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (out[(size_t)y * W + x] >= 0) continue;
            int adjacent = 0;
            const int label = labels[(size_t)y * W + x];
            out[(size_t)y * W + x] = cur;
            int size = 1, visited = 0;
            q[0] = y; q[1] = x;
            while (visited < size && size < max_size) {
                for (int i = 0; i < 4; ++i) {
                    const int yy = q[2 * visited] + ddy[i], xx = q[2 * visited + 1] + ddx[i];
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    const size_t pp = (size_t)yy * W + xx;
                    if (labels[pp] == label && out[pp] == -1) {
                        out[pp] = cur;
                        q[2 * size] = yy; q[2 * size + 1] = xx;
                        size += 1;
                        if (size >= max_size) break;
                    } else if (out[pp] >= 0 && out[pp] != cur) {
                        adjacent = out[pp];
                    }
                }
                visited += 1;
            }
            if (size < min_size) {
                for (int i = 0; i < size; ++i) out[(size_t)q[2 * i] * W + q[2 * i + 1]] = adjacent;
            } else {
                cur += 1;
            }
        }
    free(q);
    int mx = 0;
    for (size_t p = 0; p < P; ++p) if (out[p] > mx) mx = out[p];
    return mx + 1; /* n_nodes = segments.max() + 1 (graph_builder.py:158) */
}

/* =====================================================================================================================
 * The float64 path: SuperpixelGraphConfig(use_lab=False) hands `self.rgb.astype(float)` to slic (graph_builder.py:177-179).
 * A float64 input keeps every stage of skimage's slic in float64 — img_as_float leaves it alone, rgb2lab, gaussian_filter and
 * the double instance of _slic_cython's fused type (slic_superpixels.py:231-317 of 0.18.3) — so the steps above are restated
 * once more in double.  Pinned bit-exact (Gaussian, k-means, connectivity) / to 1e-12 (rgb2lab: numpy's pow and cbrt) against
 * scikit-image 0.18.3 + scipy 1.7.1 by tests/test_slic_oracle.py with tests/golden/skimage_0183_rgb.npz. */
void ggo_slic_rescale_lab64(int H, int W, const double* image, int rescale_input, double* out) {
    const size_t n = (size_t)H * W * 3;
    double mn = image[0], mx = image[0];
    for (size_t i = 1; i < n; ++i) { if (image[i] < mn) mn = image[i]; if (image[i] > mx) mx = image[i]; }
    const double range = mx - mn;
    const double m[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    for (size_t p = 0; p < (size_t)H * W; ++p) {
        double lin[3];
        for (int c = 0; c < 3; ++c) {
            double v = image[3 * p + c];
            if (rescale_input) { v = v - mn; if (mx != mn) v = v / range; }
            lin[c] = v > 0.04045 ? ggo_pow24((v + 0.055) / 1.055) : v / 12.92;
        }
        double xyz[3];
        for (int r = 0; r < 3; ++r) xyz[r] = (lin[0] * m[3 * r] + lin[1] * m[3 * r + 1]) + lin[2] * m[3 * r + 2];
        const double t[3] = {xyz[0] / 0.95047, xyz[1] / 1.0, xyz[2] / 1.08883};
        double f[3];
        for (int c = 0; c < 3; ++c) f[c] = t[c] > 0.008856 ? ggo_cbrt(t[c]) : 7.787 * t[c] + 16.0 / 116.0;
        out[3 * p + 0] = 116.0 * f[1] - 16.0;
        out[3 * p + 1] = 500.0 * (f[0] - f[1]);
        out[3 * p + 2] = 200.0 * (f[1] - f[2]);
    }
}

static void gauss_axis64(int H, int W, int C, const double* in, double* out, const double* w, int r, int axis) {
    const int len = axis == 0 ? H : W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < C; ++c) {
                const int l = axis == 0 ? y : x;
#define AT(idx) (axis == 0 ? in[((size_t)reflect_sym((idx), len) * W + x) * C + c] : in[((size_t)y * W + reflect_sym((idx), len)) * C + c])
                double tmp = AT(l) * w[r];
                for (int jj = -r; jj < 0; ++jj) tmp += (AT(l + jj) + AT(l - jj)) * w[jj + r];
#undef AT
                out[((size_t)y * W + x) * C + c] = tmp;
            }
}

void ggo_gaussian_f64(int H, int W, int C, const double* in, double sigma, double* out) {
    double w[129];
    const int r = ggo_gaussian_weights(sigma, w, 129);
    double* tmp = (double*)malloc((size_t)H * W * C * sizeof(double));
    /* slic filters the (1, H, W, C) volume along its depth axis first: every tap of a length-1 reflected axis is the pixel
     * itself, v w_r + sum (v + v) w_k — not exactly v in float64 (the float32 path rounds the difference away) */
    for (size_t i = 0; i < (size_t)H * W * C; ++i) {
        double t = in[i] * w[r];
        for (int jj = -r; jj < 0; ++jj) t += (in[i] + in[i]) * w[jj + r];
        out[i] = t;
    }
    gauss_axis64(H, W, C, out, tmp, w, r, 0);
    gauss_axis64(H, W, C, tmp, out, w, r, 1);
    free(tmp);
}

/* _slic_cython, double instance: ggo_slic_kmeans with every float a double */
void ggo_slic_kmeans64(int H, int W, const double* image, int K, double* centers, double step, int max_iter, int32_t* labels) {
    int wy, wx, d0, d1, d2, d3;
    ggo_slic_grid(H, W, K, &wy, &wx, &d0, &d1, &d2, &d3);
    const double sw = 1.0 / (step * step);
    const size_t P = (size_t)H * W;
    double* dist = (double*)malloc(P * sizeof(double));
    int64_t* cnt = (int64_t*)malloc((size_t)K * sizeof(int64_t));
    for (size_t p = 0; p < P; ++p) labels[p] = 0;
    for (int it = 0; it < max_iter; ++it) {
        int change = 0;
        for (size_t p = 0; p < P; ++p) dist[p] = INFINITY;
        for (int k = 0; k < K; ++k) {
            const double cy = centers[5 * k + 0], cx = centers[5 * k + 1];
            if (cy != cy || cx != cx) continue;
            double fy0 = cy - (double)(2 * wy); if (!(fy0 > 0.0)) fy0 = 0.0;
            double fy1 = cy + (double)(2 * wy) + 1.0; if (!(fy1 < (double)H)) fy1 = (double)H;
            double fx0 = cx - (double)(2 * wx); if (!(fx0 > 0.0)) fx0 = 0.0;
            double fx1 = cx + (double)(2 * wx) + 1.0; if (!(fx1 < (double)W)) fx1 = (double)W;
            const int y0 = (int)fy0, y1 = (int)fy1, x0 = (int)fx0, x1 = (int)fx1;
            const double c0 = centers[5 * k + 2], c1 = centers[5 * k + 3], c2 = centers[5 * k + 4];
            for (int y = y0; y < y1; ++y) {
                const double ty = cy - (double)y;
                const double dy = ty * ty;
                for (int x = x0; x < x1; ++x) {
                    const double tx = cx - (double)x;
                    double d = (dy + tx * tx) * sw;
                    const double* px = image + ((size_t)y * W + x) * 3;
                    double dc = 0.0, t;
                    t = px[0] - c0; dc += t * t;
                    t = px[1] - c1; dc += t * t;
                    t = px[2] - c2; dc += t * t;
                    d += dc;
                    const size_t p = (size_t)y * W + x;
                    if (dist[p] > d) { labels[p] = k; dist[p] = d; change = 1; }
                }
            }
        }
        if (!change) break;
        memset(cnt, 0, (size_t)K * sizeof(int64_t));
        for (int k = 0; k < 5 * K; ++k) centers[k] = 0.0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t p = (size_t)y * W + x;
                const int k = labels[p];
                cnt[k] += 1;
                centers[5 * k + 0] += (double)y;
                centers[5 * k + 1] += (double)x;
                centers[5 * k + 2] += image[3 * p + 0];
                centers[5 * k + 3] += image[3 * p + 1];
                centers[5 * k + 4] += image[3 * p + 2];
            }
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 5; ++c) centers[5 * k + c] = centers[5 * k + c] / (double)cnt[k];
    }
    free(dist); free(cnt);
}

/* the whole call for use_lab=False: bgr uint8 -> rgb.astype(float) -> slic */
int ggo_slic_rgb(int H, int W, const uint8_t* bgr, int n_segments, double compactness, double sigma, int32_t* segments) {
    const size_t P = (size_t)H * W;
    double* a = (double*)malloc(P * 3 * sizeof(double));
    double* b = (double*)malloc(P * 3 * sizeof(double));
    for (size_t p = 0; p < P; ++p) { a[3 * p] = (double)bgr[3 * p + 2]; a[3 * p + 1] = (double)bgr[3 * p + 1]; a[3 * p + 2] = (double)bgr[3 * p]; }
    ggo_slic_rescale_lab64(H, W, a, 1, b);
    int sy, sx, y0, x0, ny, nx;
    const int K = ggo_slic_grid(H, W, n_segments, &sy, &sx, &y0, &x0, &ny, &nx);
    if (sigma > 0.0) ggo_gaussian_f64(H, W, 3, b, sigma, a);
    else memcpy(a, b, P * 3 * sizeof(double));
    const double ratio = 1.0 / compactness;
    for (size_t i = 0; i < P * 3; ++i) a[i] = a[i] * ratio;
    double* centers = (double*)calloc((size_t)K * 5, sizeof(double));
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            centers[5 * (j * nx + i) + 0] = (double)(y0 + j * sy);
            centers[5 * (j * nx + i) + 1] = (double)(x0 + i * sx);
        }
    const double step = (double)(sy > sx ? sy : sx);
    int32_t* raw = (int32_t*)malloc(P * sizeof(int32_t));
    ggo_slic_kmeans64(H, W, a, K, centers, step, 10, raw);
    const double seg_size = (double)P / (double)K;
    const int n = ggo_slic_connectivity(H, W, raw, (int)(0.5 * seg_size), (int)(3.0 * seg_size), segments);
    free(a); free(b); free(centers); free(raw);
    return n;
}

/* ---- the whole call */
int ggo_slic(int H, int W, const float* image, int n_segments, float compactness,
             float sigma, int rescale_input, int32_t* segments) {
    const size_t P = (size_t)H * W;
    float* a = (float*)malloc(P * 3 * sizeof(float));
    float* b = (float*)malloc(P * 3 * sizeof(float));
    ggo_slic_rescale_lab(H, W, image, rescale_input, a);
    int sy, sx, y0, x0, ny, nx;
    const int K = ggo_slic_grid(H, W, n_segments, &sy, &sx, &y0, &x0, &ny, &nx);
    if (sigma > 0.0f) ggo_gaussian_f32(H, W, 3, a, (double)sigma, b);
    else memcpy(b, a, P * 3 * sizeof(float));
    const float ratio = (float)(1.0 / (double)compactness);
    for (size_t i = 0; i < P * 3; ++i) b[i] = b[i] * ratio;
    float* centers = (float*)calloc((size_t)K * 5, sizeof(float));
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            centers[5 * (j * nx + i) + 0] = (float)(y0 + j * sy);
            centers[5 * (j * nx + i) + 1] = (float)(x0 + i * sx);
        }
    const float step = (float)(sy > sx ? sy : sx); /* step = max(steps); z step is 1 */
    int32_t* raw = (int32_t*)malloc(P * sizeof(int32_t));
    ggo_slic_kmeans(H, W, b, K, centers, step, 10, raw);
    const double seg_size = (double)P / (double)K;
    const int n = ggo_slic_connectivity(H, W, raw, (int)(0.5 * seg_size), (int)(3.0 * seg_size), segments);
    free(a); free(b); free(centers); free(raw);
    return n;
}
