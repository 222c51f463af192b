/*
 * oracle/graph.c — region statistics, node features, RAG + non-local edges,
 * automatic prior.  TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Follows reference src/gcn_grabcut/graph_builder.py:
 *   _region_statistics :190-226, _assemble_node_features :228-255,
 *   _compute_edges :257-307, _pair_features :309-322, _nonlocal_pairs :324-350,
 *   compute_auto_prior :357-444, _unit_norm :447-454
 * with numpy's arithmetic restated operation by operation: np.bincount(weights)
 * accumulates in float64 in raster order and is then cast to float32; array
 * expressions are float32 with python-float scalars cast to float32; reductions
 * along the contiguous axis use numpy's pairwise summation; find_boundaries
 * (mode="inner") is skimage's.
 * np.argpartition's tie order is unspecified: here "k smallest, ties to the
 * lowest index" (SURVEY hard part 6).
 */
#include "ggc_oracle.h"
#include "../include/ggc_fmath.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct ggo_graph {
    int N, n_pairs;
    float* feat;      /* [N,16] */
    float* prior;     /* [N,3]  */
    float* cent;      /* [N,2]  */
    float* area;      /* [N]    */
    int32_t* pairs;   /* [n_pairs,2] lo,hi: adjacency sorted then non-local sorted */
    float* attr;      /* [n_pairs,5] */
};

/* numpy pairwise summation, float32 accumulators */
static float pairwise_f32(const float* a, int n) {
    if (n < 8) { float s = 0.0f; for (int i = 0; i < n; ++i) s += a[i]; return s; }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_f32(a, n2) + pairwise_f32(a + n2, n - n2);
}

/* skimage.segmentation.find_boundaries(seg, mode="inner"), connectivity 1 */
void ggo_find_boundaries_inner(int H, int W, const int32_t* seg, uint8_t* out) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int32_t v = seg[(size_t)y * W + x];
            int32_t mx = v, mn = v;
            if (y > 0)     { int32_t u = seg[(size_t)(y - 1) * W + x]; if (u > mx) mx = u; if (u < mn) mn = u; }
            if (y < H - 1) { int32_t u = seg[(size_t)(y + 1) * W + x]; if (u > mx) mx = u; if (u < mn) mn = u; }
            if (x > 0)     { int32_t u = seg[(size_t)y * W + x - 1]; if (u > mx) mx = u; if (u < mn) mn = u; }
            if (x < W - 1) { int32_t u = seg[(size_t)y * W + x + 1]; if (u > mx) mx = u; if (u < mn) mn = u; }
            out[(size_t)y * W + x] = (uint8_t)((mx != mn) && (v != 0));
        }
}

static int cmp_i64(const void* a, const void* b) {
    const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}

static float norm3(const float* a, const float* b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return sqrtf((dx * dx + dy * dy) + dz * dz);
}

/* _pair_features (graph_builder.py:309-322) */
static void pair_features(int n, const int32_t* pairs, const float* mean_lab, const float* cent,
                          const float* mgn, const float* shared, float flag, float* attr) {
    float* de = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    float* dx = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    float de_max = -INFINITY, dx_max = -INFINITY;
    for (int e = 0; e < n; ++e) {
        const int i = pairs[2 * e], j = pairs[2 * e + 1];
        de[e] = norm3(mean_lab + 3 * i, mean_lab + 3 * j);
        const float a = cent[2 * i] - cent[2 * j], b = cent[2 * i + 1] - cent[2 * j + 1];
        dx[e] = sqrtf(a * a + b * b);
        if (de[e] > de_max) de_max = de[e];
        if (dx[e] > dx_max) dx_max = dx[e];
    }
    const float de_den = (float)((double)de_max + 1e-6), dx_den = (float)((double)dx_max + 1e-6);
    for (int e = 0; e < n; ++e) {
        const int i = pairs[2 * e], j = pairs[2 * e + 1];
        attr[5 * e + 0] = de[e] / de_den;
        attr[5 * e + 1] = dx[e] / dx_den;
        attr[5 * e + 2] = shared ? shared[e] : 0.0f;
        attr[5 * e + 3] = fabsf(mgn[i] - mgn[j]);
        attr[5 * e + 4] = flag;
    }
    free(de); free(dx);
}

static void unit_norm(float* v, int n) { /* _unit_norm (graph_builder.py:447-454) */
    float mn = v[0], mx = v[0];
    for (int i = 1; i < n; ++i) { if (v[i] < mn) mn = v[i]; if (v[i] > mx) mx = v[i]; }
    if ((double)mx - (double)mn < 1e-8) { for (int i = 0; i < n; ++i) v[i] = 0.0f; return; }
    const float den = (float)((double)mx - (double)mn);
    for (int i = 0; i < n; ++i) v[i] = (v[i] - mn) / den;
}

static float fix_nan(float v, float posinf) { /* np.nan_to_num(nan=0, posinf=1, neginf=0) */
    if (v != v) return 0.0f;
    if (isinf(v)) return v > 0 ? posinf : 0.0f;
    return v;
}

/* compute_auto_prior (graph_builder.py:357-444) */
void ggo_auto_prior_sigmas(int H, int W, const int32_t* seg, const float* lab, int N, double centre_sigma, double contrast_sigma,
                           float* prior) {
    const size_t P = (size_t)H * W;
    double* acc = (double*)calloc((size_t)N * 6, sizeof(double)); /* cnt, l, a, b, y, x */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            double* a = acc + (size_t)seg[p] * 6;
            a[0] += 1.0;
            a[1] += (double)lab[3 * p]; a[2] += (double)lab[3 * p + 1]; a[3] += (double)lab[3 * p + 2];
            a[4] += (double)y / (double)H; a[5] += (double)x / (double)W;
        }
    float* counts = (float*)malloc((size_t)N * sizeof(float));
    float* safe = (float*)malloc((size_t)N * sizeof(float));
    float* ml = (float*)malloc((size_t)N * 3 * sizeof(float));
    float* ce = (float*)malloc((size_t)N * 2 * sizeof(float));
    for (int i = 0; i < N; ++i) {
        counts[i] = (float)acc[6 * i];
        safe[i] = counts[i] > 1.0f ? counts[i] : 1.0f;
        for (int c = 0; c < 3; ++c) ml[3 * i + c] = (float)acc[6 * i + 1 + c] / safe[i];
        ce[2 * i + 0] = (float)(acc[6 * i + 4] / (double)safe[i]);
        ce[2 * i + 1] = (float)(acc[6 * i + 5] / (double)safe[i]);
    }
    /* Summation orders.  The reference sums with numpy (pairwise float32 sums); a float sum has no canonical order, and the
     * restatement fixes THE ORDER OF THE MI355X KERNELS (k_contrast / k_prior in csrc/ggc_graph.hip), so that the two produce
     * the same bits and a trimap can be compared pixel for pixel end to end: a lane (of 64) or a thread (of 256) sums every
     * 64th / 256th term in index order, then the partial sums are combined by the butterfly / halving tree written out below.
     * exp is the fixed IEEE sequence of include/ggc_fmath.h on both sides.  Against the reference's own output the result
     * stays within 2e-5 (tests/test_graph_oracle.py). */
    /* cue 1: spatially weighted global colour contrast */
    float csum = (float)((double)H * (double)W);          /* counts.sum(): exact in float32 below 2^24 pixels */
    if (!(csum > 1.0f)) csum = 1.0f;
    float* contrast = (float*)malloc((size_t)N * sizeof(float));
    float* row = NULL;
    const float two_cs2 = (float)(2 * contrast_sigma * contrast_sigma);   /* python float, then a float32 operand */
    for (int i = 0; i < N; ++i) {
        float lane[64];
        for (int l = 0; l < 64; ++l) {
            float acc = 0.0f;
            for (int j = l; j < N; j += 64) {
                const float cd = norm3(ml + 3 * i, ml + 3 * j);
                const float a = ce[2 * i] - ce[2 * j], b = ce[2 * i + 1] - ce[2 * j + 1];
                const float sd = sqrtf(a * a + b * b);
                acc += (cd * ggc_expf(-(sd * sd) / two_cs2)) * (counts[j] / csum);
            }
            lane[l] = acc;
        }
        for (int o = 32; o > 0; o >>= 1) {                 /* wave butterfly: every lane ends with the same sum */
            float t[64];
            for (int l = 0; l < 64; ++l) t[l] = lane[l] + lane[l ^ o];
            memcpy(lane, t, sizeof t);
        }
        contrast[i] = lane[0];
    }
    unit_norm(contrast, N);
    const float two_ce2 = (float)(2 * centre_sigma * centre_sigma);
    float* fg = (float*)malloc((size_t)N * sizeof(float));
    for (int i = 0; i < N; ++i) {
        const float a = ce[2 * i] - 0.5f, b = ce[2 * i + 1] - 0.5f;
        const float d = sqrtf(a * a + b * b);
        fg[i] = contrast[i] * ggc_expf(-(d * d) / two_ce2);
    }
    unit_norm(fg, N);
    /* cue 2: background model from the image frame */
    float* bc = (float*)calloc((size_t)N, sizeof(float));
    for (int x = 0; x < W; ++x) { bc[seg[x]] += 1.0f; }
    for (int x = 0; x < W; ++x) { bc[seg[(size_t)(H - 1) * W + x]] += 1.0f; }
    for (int y = 0; y < H; ++y) { bc[seg[(size_t)y * W]] += 1.0f; }
    for (int y = 0; y < H; ++y) { bc[seg[(size_t)y * W + W - 1]] += 1.0f; }
    float* bg = (float*)calloc((size_t)N, sizeof(float));
    {
        /* weighted mean / variance of the frame regions' colours: double partial sums of 256 strided threads, halving tree */
        double ds[4][256];
        for (int t = 0; t < 256; ++t) {
            double bs = 0, m0 = 0, m1 = 0, m2 = 0;
            for (int i = t; i < N; i += 256) {
                const double w = (double)bc[i];
                bs += w; m0 += w * (double)ml[3 * i]; m1 += w * (double)ml[3 * i + 1]; m2 += w * (double)ml[3 * i + 2];
            }
            ds[0][t] = bs; ds[1][t] = m0; ds[2][t] = m1; ds[3][t] = m2;
        }
        for (int o = 128; o > 0; o >>= 1) for (int t = 0; t < o; ++t) for (int c = 0; c < 4; ++c) ds[c][t] += ds[c][t + o];
        const float bsum = (float)ds[0][0];
        const float mu[3] = {(float)(ds[1][0] / ds[0][0]), (float)(ds[2][0] / ds[0][0]), (float)(ds[3][0] / ds[0][0])};
        for (int t = 0; t < 256; ++t) {
            double var = 0;
            for (int i = t; i < N; i += 256) {
                const float w = bc[i] / bsum;
                const float d0 = ml[3 * i] - mu[0], d1 = ml[3 * i + 1] - mu[1], d2 = ml[3 * i + 2] - mu[2];
                var += (double)((d0 * d0) * w) + (double)((d1 * d1) * w) + (double)((d2 * d2) * w);
            }
            ds[0][t] = var;
        }
        for (int o = 128; o > 0; o >>= 1) for (int t = 0; t < o; ++t) ds[0][t] += ds[0][t + o];
        const float var_bg = (float)ds[0][0];
        const double sigma_bg = var_bg > 1e-6 ? (double)sqrtf(var_bg) : sqrt(1e-6);
        const float den = (float)(2.0 * (sigma_bg + 1e-6) * (sigma_bg + 1e-6));
        for (int i = 0; i < N; ++i) {
            float v = 0.0f;
            if (bsum > 0.0f) {
                const float d0 = ml[3 * i] - mu[0], d1 = ml[3 * i + 1] - mu[1], d2 = ml[3 * i + 2] - mu[2];
                const float dd = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
                v = ggc_expf(-(dd * dd) / den);
            }
            float r = (bc[i] / safe[i]) * 4.0f;
            r = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            bg[i] = (v != v) ? v : (v > r ? v : r);        /* np.maximum propagates NaN */
        }
    }
    unit_norm(bg, N);
    for (int i = 0; i < N; ++i) {
        prior[3 * i + 0] = fix_nan(fg[i], 1.0f);
        prior[3 * i + 1] = fix_nan(bg[i], 1.0f);
        prior[3 * i + 2] = fix_nan(1.0f - fabsf(fg[i] - bg[i]), 1.0f);
    }
    (void)P;
    free(acc); free(counts); free(safe); free(ml); free(ce); free(contrast); free(row); free(fg); free(bc); free(bg);
}

void ggo_auto_prior(int H, int W, const int32_t* seg, const float* lab, int N, float* prior) {   /* the reference's defaults (:357-362) */
    ggo_auto_prior_sigmas(H, W, seg, lab, N, 0.45, 0.40, prior);
}

ggo_graph* ggo_graph_build(int H, int W, const int32_t* seg, const float* lab, const float* hsv,
                           const float* grad, int connectivity, int n_nonlocal, int* n_nodes, int* n_edges) {
    const size_t P = (size_t)H * W;
    int N = 0;
    for (size_t p = 0; p < P; ++p) if (seg[p] + 1 > N) N = seg[p] + 1;
    ggo_graph* g = (ggo_graph*)calloc(1, sizeof(ggo_graph));
    g->N = N;

    /* ---- _region_statistics: 14 float64 bincount passes in raster order */
    enum { S_CNT, S_L, S_A, S_B, S_L2, S_A2, S_B2, S_H, S_S, S_V, S_Y, S_X, S_BND, S_G, S_GN, S_NUM };
    double* acc = (double*)calloc((size_t)N * S_NUM, sizeof(double));
    uint8_t* bnd = (uint8_t*)malloc(P);
    ggo_find_boundaries_inner(H, W, seg, bnd);
    float gmax = grad[0];
    for (size_t p = 1; p < P; ++p) if (grad[p] > gmax) gmax = grad[p];
    const float gden = (float)((double)gmax + 1e-6);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            double* a = acc + (size_t)seg[p] * S_NUM;
            a[S_CNT] += 1.0;
            for (int c = 0; c < 3; ++c) {
                const float v = lab[3 * p + c];
                a[S_L + c] += (double)v;
                a[S_L2 + c] += (double)(v * v);
                a[S_H + c] += (double)hsv[3 * p + c];
            }
            a[S_Y] += (double)((float)y / (float)H);
            a[S_X] += (double)((float)x / (float)W);
            a[S_BND] += (double)bnd[p];
            a[S_G] += (double)grad[p];
            a[S_GN] += (double)(grad[p] / gden);
        }
    float* counts = (float*)malloc((size_t)N * sizeof(float));
    float* safe = (float*)malloc((size_t)N * sizeof(float));
    float* mean_lab = (float*)malloc((size_t)N * 3 * sizeof(float));
    float* std_lab = (float*)malloc((size_t)N * 3 * sizeof(float));
    float* mean_hsv = (float*)malloc((size_t)N * 3 * sizeof(float));
    float* bpx = (float*)malloc((size_t)N * sizeof(float));
    float* mgrad = (float*)malloc((size_t)N * sizeof(float));
    float* mgn = (float*)malloc((size_t)N * sizeof(float));
    g->cent = (float*)malloc((size_t)N * 2 * sizeof(float));
    g->area = (float*)malloc((size_t)N * sizeof(float));
    for (int i = 0; i < N; ++i) {
        const double* a = acc + (size_t)i * S_NUM;
        counts[i] = (float)a[S_CNT];
        safe[i] = counts[i] > 1.0f ? counts[i] : 1.0f;
        for (int c = 0; c < 3; ++c) {
            const float m = (float)a[S_L + c] / safe[i];
            const float sq = (float)a[S_L2 + c] / safe[i];
            float v = sq - m * m;
            if (!(v > 0.0f)) v = (v != v) ? v : 0.0f; /* np.maximum(., 0) propagates NaN */
            mean_lab[3 * i + c] = m;
            std_lab[3 * i + c] = sqrtf(v);
            mean_hsv[3 * i + c] = (float)a[S_H + c] / safe[i];
        }
        g->cent[2 * i + 0] = (float)a[S_Y] / safe[i];
        g->cent[2 * i + 1] = (float)a[S_X] / safe[i];
        bpx[i] = (float)a[S_BND];
        mgrad[i] = (float)a[S_G] / safe[i];
        mgn[i] = (float)a[S_GN] / safe[i];
        g->area[i] = counts[i] / (float)((double)H * (double)W);
    }

    /* ---- _assemble_node_features */
    g->feat = (float*)calloc((size_t)N * 16, sizeof(float));
    const float four_pi = (float)(4 * 3.141592653589793);
    for (int i = 0; i < N; ++i) {
        float* f = g->feat + (size_t)i * 16;
        for (int c = 0; c < 3; ++c) { f[c] = mean_lab[3 * i + c]; f[3 + c] = std_lab[3 * i + c]; f[6 + c] = mean_hsv[3 * i + c]; }
        f[9] = g->cent[2 * i]; f[10] = g->cent[2 * i + 1];
        f[11] = g->area[i];
        const float per = bpx[i] > 1.0f ? bpx[i] : 1.0f;
        float comp = (four_pi * counts[i]) / (per * per);
        comp = comp < 0.0f ? 0.0f : (comp > 1.0f ? 1.0f : comp);
        f[12] = comp;
        f[13] = mgrad[i] / (float)255.0;
        f[14] = bpx[i] / safe[i];
        const float a = g->cent[2 * i] - 0.5f, b = g->cent[2 * i + 1] - 0.5f;
        f[15] = sqrtf(a * a + b * b) / (float)0.707;
    }
    for (int c = 0; c < 6; ++c) { /* per-image min-max of the colour statistics (:250-253) */
        float mn = g->feat[c], mx = g->feat[c];
        for (int i = 1; i < N; ++i) { const float v = g->feat[(size_t)i * 16 + c]; if (v < mn) mn = v; if (v > mx) mx = v; }
        const float den = (mx - mn) + (float)1e-6;
        for (int i = 0; i < N; ++i) g->feat[(size_t)i * 16 + c] = (g->feat[(size_t)i * 16 + c] - mn) / den;
    }
    for (size_t t = 0; t < (size_t)N * 16; ++t) g->feat[t] = fix_nan(g->feat[t], 1.0f);

    /* ---- _compute_edges: adjacency by shifted comparison + np.unique(return_counts) */
    size_t cap = 2 * P + 16;
    if (connectivity == 8) cap += 2 * P;
    int64_t* codes = (int64_t*)malloc(cap * sizeof(int64_t));
    size_t nc = 0;
#define PUSH(a_, b_) do { int32_t a__ = (a_), b__ = (b_); if (a__ != b__) { \
        int64_t lo = a__ < b__ ? a__ : b__, hi = a__ < b__ ? b__ : a__; codes[nc++] = lo * N + hi; } } while (0)
    for (int y = 0; y < H; ++y) for (int x = 0; x + 1 < W; ++x) PUSH(seg[(size_t)y * W + x], seg[(size_t)y * W + x + 1]);
    for (int y = 0; y + 1 < H; ++y) for (int x = 0; x < W; ++x) PUSH(seg[(size_t)y * W + x], seg[(size_t)(y + 1) * W + x]);
    if (connectivity == 8) {
        for (int y = 0; y + 1 < H; ++y) for (int x = 0; x + 1 < W; ++x) PUSH(seg[(size_t)y * W + x], seg[(size_t)(y + 1) * W + x + 1]);
        for (int y = 0; y + 1 < H; ++y) for (int x = 0; x + 1 < W; ++x) PUSH(seg[(size_t)y * W + x + 1], seg[(size_t)(y + 1) * W + x]);
    }
#undef PUSH
    qsort(codes, nc, sizeof(int64_t), cmp_i64);
    int n_adj = 0;
    for (size_t i = 0; i < nc; ++i) if (i == 0 || codes[i] != codes[i - 1]) ++n_adj;
    int32_t* adj = (int32_t*)malloc((size_t)(n_adj > 0 ? n_adj : 1) * 2 * sizeof(int32_t));
    float* shared = (float*)malloc((size_t)(n_adj > 0 ? n_adj : 1) * sizeof(float));
    int64_t smax = 0;
    {
        int k = -1;
        int64_t run = 0;
        for (size_t i = 0; i < nc; ++i) {
            if (i == 0 || codes[i] != codes[i - 1]) {
                if (k >= 0) { shared[k] = (float)run; if (run > smax) smax = run; }
                ++k; run = 0;
                adj[2 * k] = (int32_t)(codes[i] / N); adj[2 * k + 1] = (int32_t)(codes[i] % N);
            }
            ++run;
        }
        if (k >= 0) { shared[k] = (float)run; if (run > smax) smax = run; }
    }
    const float sden = (float)((double)smax + 1e-6);
    for (int e = 0; e < n_adj; ++e) shared[e] = shared[e] / sden;
    float* attr_adj = (float*)malloc((size_t)(n_adj > 0 ? n_adj : 1) * 5 * sizeof(float));
    pair_features(n_adj, adj, mean_lab, g->cent, mgn, shared, 0.0f, attr_adj);

    /* ---- _nonlocal_pairs: k nearest in mean-Lab, excluding self and adjacent */
    int n_nl = 0;
    int32_t* nl = NULL;
    float* attr_nl = NULL;
    if (n_nonlocal > 0 && N > n_nonlocal + 1) {
        const int k = n_nonlocal;
        uint8_t* isadj = (uint8_t*)calloc((size_t)N * N, 1);
        for (int e = 0; e < n_adj; ++e) { isadj[(size_t)adj[2 * e] * N + adj[2 * e + 1]] = 1; isadj[(size_t)adj[2 * e + 1] * N + adj[2 * e]] = 1; }
        int64_t* nlc = (int64_t*)malloc((size_t)N * k * sizeof(int64_t));
        size_t nn = 0;
        float* d = (float*)malloc((size_t)N * sizeof(float));
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j)
                d[j] = (j == i || isadj[(size_t)i * N + j]) ? INFINITY : norm3(mean_lab + 3 * i, mean_lab + 3 * j);
            for (int t = 0; t < k; ++t) { /* k smallest, ties to the lowest index */
                int best = -1;
                for (int j = 0; j < N; ++j) if (d[j] == d[j] && (best < 0 || d[j] < d[best])) best = j;
                if (best < 0) break;
                const float dv = d[best];
                d[best] = NAN; /* taken */
                if (!isfinite(dv)) continue;
                const int64_t lo = i < best ? i : best, hi = i < best ? best : i;
                nlc[nn++] = lo * N + hi;
            }
        }
        qsort(nlc, nn, sizeof(int64_t), cmp_i64);
        for (size_t i = 0; i < nn; ++i) if (i == 0 || nlc[i] != nlc[i - 1]) ++n_nl;
        nl = (int32_t*)malloc((size_t)(n_nl > 0 ? n_nl : 1) * 2 * sizeof(int32_t));
        int q = 0;
        for (size_t i = 0; i < nn; ++i) if (i == 0 || nlc[i] != nlc[i - 1]) { nl[2 * q] = (int32_t)(nlc[i] / N); nl[2 * q + 1] = (int32_t)(nlc[i] % N); ++q; }
        attr_nl = (float*)malloc((size_t)(n_nl > 0 ? n_nl : 1) * 5 * sizeof(float));
        if (n_nl) pair_features(n_nl, nl, mean_lab, g->cent, mgn, NULL, 1.0f, attr_nl);
        free(isadj); free(nlc); free(d);
    }
    g->n_pairs = n_adj + n_nl;
    g->pairs = (int32_t*)malloc((size_t)(g->n_pairs > 0 ? g->n_pairs : 1) * 2 * sizeof(int32_t));
    g->attr = (float*)malloc((size_t)(g->n_pairs > 0 ? g->n_pairs : 1) * 5 * sizeof(float));
    memcpy(g->pairs, adj, (size_t)n_adj * 2 * sizeof(int32_t));
    memcpy(g->attr, attr_adj, (size_t)n_adj * 5 * sizeof(float));
    if (n_nl) {
        memcpy(g->pairs + 2 * n_adj, nl, (size_t)n_nl * 2 * sizeof(int32_t));
        memcpy(g->attr + 5 * n_adj, attr_nl, (size_t)n_nl * 5 * sizeof(float));
    }
    g->prior = (float*)malloc((size_t)N * 3 * sizeof(float));
    ggo_auto_prior(H, W, seg, lab, N, g->prior);

    if (n_nodes) *n_nodes = N;
    if (n_edges) *n_edges = 2 * g->n_pairs;
    free(acc); free(bnd); free(counts); free(safe); free(mean_lab); free(std_lab); free(mean_hsv);
    free(bpx); free(mgrad); free(mgn); free(codes); free(adj); free(shared); free(attr_adj); free(nl); free(attr_nl);
    return g;
}

void ggo_graph_get(const ggo_graph* g, float* feat, float* prior, float* cent, float* area,
                   int64_t* edge_index, float* edge_attr) {
    const int N = g->N, E2 = g->n_pairs;
    if (feat) memcpy(feat, g->feat, (size_t)N * 16 * sizeof(float));
    if (prior) memcpy(prior, g->prior, (size_t)N * 3 * sizeof(float));
    if (cent) memcpy(cent, g->cent, (size_t)N * 2 * sizeof(float));
    if (area) memcpy(area, g->area, (size_t)N * sizeof(float));
    if (edge_index) { /* src = [lo.., hi..], dst = [hi.., lo..] (graph_builder.py:303-306) */
        for (int e = 0; e < E2; ++e) {
            edge_index[e] = g->pairs[2 * e];            edge_index[E2 + e] = g->pairs[2 * e + 1];
            edge_index[2 * E2 + e] = g->pairs[2 * e + 1]; edge_index[2 * E2 + E2 + e] = g->pairs[2 * e];
        }
    }
    if (edge_attr) {
        memcpy(edge_attr, g->attr, (size_t)E2 * 5 * sizeof(float));
        memcpy(edge_attr + (size_t)E2 * 5, g->attr, (size_t)E2 * 5 * sizeof(float));
    }
}

void ggo_graph_free(ggo_graph* g) {
    if (!g) return;
    free(g->feat); free(g->prior); free(g->cent); free(g->area); free(g->pairs); free(g->attr);
    free(g);
}
