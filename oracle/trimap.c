/*
 * oracle/trimap.c — region probabilities -> pixel trimap.
 * TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Follows reference src/gcn_grabcut/pipeline.py: guided_filter :71-100,
 * refine_trimap :103-146, _seed_from_prior :149-186, and model.py:
 * probs_to_node_trimap :623-645, project_to_pixels :648-661, _probs_to_trimap
 * :664-678.
 *
 * cv2.blur is absent here (PARITY UNPINNED): restated from SURVEY.md Appendix
 * A.4 as a normalised (2r+1)^2 box with BORDER_REFLECT_101, float64 sums in a
 * fixed order — the 2r+1 taps of a row left to right, then the 2r+1 row sums
 * top to bottom — times 1/(2r+1)^2, cast to float32.  Everything else is
 * float32 elementwise, exactly as numpy evaluates the reference expressions.
 * np.argsort's order among equal keys is unspecified; here ascending-stable
 * reversed (among ties the larger index ranks first).
 */
#include "ggc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int refl101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; }
    return i;
}

void ggo_box_blur(int H, int W, const float* in, int radius, float* out) {
    const int k = 2 * radius + 1;
    const double scale = 1.0 / ((double)k * (double)k);
    double* hs = (double*)malloc((size_t)H * W * sizeof(double));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double s = 0.0;
            for (int d = -radius; d <= radius; ++d) s += (double)in[(size_t)y * W + refl101(x + d, W)];
            hs[(size_t)y * W + x] = s;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double s = 0.0;
            for (int d = -radius; d <= radius; ++d) s += hs[(size_t)refl101(y + d, H) * W + x];
            out[(size_t)y * W + x] = (float)(s * scale);
        }
    free(hs);
}

void ggo_guided_filter(int H, int W, const float* guide, const float* src, int radius, float eps, float* out) {
    const size_t P = (size_t)H * W;
    float* buf = (float*)malloc(P * 8 * sizeof(float));
    float *mg = buf, *ms = buf + P, *gs = buf + 2 * P, *gg = buf + 3 * P, *a = buf + 4 * P, *b = buf + 5 * P,
          *t0 = buf + 6 * P, *t1 = buf + 7 * P;
    ggo_box_blur(H, W, guide, radius, mg);
    ggo_box_blur(H, W, src, radius, ms);
    for (size_t p = 0; p < P; ++p) { t0[p] = guide[p] * src[p]; t1[p] = guide[p] * guide[p]; }
    ggo_box_blur(H, W, t0, radius, gs);
    ggo_box_blur(H, W, t1, radius, gg);
    for (size_t p = 0; p < P; ++p) {
        const float cov = gs[p] - mg[p] * ms[p];
        const float var = gg[p] - mg[p] * mg[p];
        a[p] = cov / (var + eps);
        b[p] = ms[p] - a[p] * mg[p];
    }
    ggo_box_blur(H, W, a, radius, t0);
    ggo_box_blur(H, W, b, radius, t1);
    for (size_t p = 0; p < P; ++p) out[p] = t0[p] * guide[p] + t1[p];
    free(buf);
}

void ggo_refine_trimap(int H, int W, const float* probs, int n_probs, const int32_t* seg,
                       const uint8_t* bgr, float thr_fg, float thr_bg, int radius, float eps,
                       int edge_aware, uint8_t* trimap) {
    const size_t P = (size_t)H * W;
    if (!edge_aware) { /* _probs_to_trimap: per-node labels, PR_BGD padding, gather */
        for (size_t p = 0; p < P; ++p) {
            const int s = seg[p];
            uint8_t lab = 2;
            if (s < n_probs) {
                const float bg = probs[3 * s + 0], fg = probs[3 * s + 2];
                lab = fg > bg ? 3 : 2;
                if (bg >= thr_bg) lab = 0;
                if (fg >= thr_fg) lab = 1;
            }
            trimap[p] = lab;
        }
        return;
    }
    float* buf = (float*)malloc(P * 5 * sizeof(float));
    float *guide = buf, *pbg = buf + P, *pfg = buf + 2 * P, *qbg = buf + 3 * P, *qfg = buf + 4 * P;
    for (size_t p = 0; p < P; ++p) {
        const int g8 = (bgr[3 * p] * 3735 + bgr[3 * p + 1] * 19235 + bgr[3 * p + 2] * 9798 + (1 << 14)) >> 15;
        guide[p] = (float)g8 / (float)255.0;
        const int s = seg[p];
        pbg[p] = s < n_probs ? probs[3 * s + 0] : 0.0f;   /* project_to_pixels zero-pads */
        pfg[p] = s < n_probs ? probs[3 * s + 2] : 0.0f;
    }
    ggo_guided_filter(H, W, guide, pbg, radius, eps, qbg);
    ggo_guided_filter(H, W, guide, pfg, radius, eps, qfg);
    for (size_t p = 0; p < P; ++p) {
        float b = qbg[p], f = qfg[p];
        b = b < 0.0f ? 0.0f : (b > 1.0f ? 1.0f : b);     /* np.clip; NaN passes through */
        f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
        uint8_t lab = f > b ? 3 : 2;
        if (b >= thr_bg) lab = 0;
        if (f >= thr_fg) lab = 1;                         /* FG wins (pipeline.py:143-145) */
        trimap[p] = lab;
    }
    free(buf);
}

void ggo_seed_from_prior(int H, int W, const float* prior, int n_nodes, const int32_t* seg,
                         double seed_frac, uint8_t* trimap) {
    const size_t P = (size_t)H * W;
    int has_fg = 0, has_bg = 0;
    for (size_t p = 0; p < P; ++p) { if (trimap[p] == 1 || trimap[p] == 3) has_fg = 1; else has_bg = 1; }
    if ((has_fg && has_bg) || n_nodes <= 0) return;
    int n_seed = (int)rint(seed_frac * (double)n_nodes);   /* python round(): half to even */
    if (n_seed < 1) n_seed = 1;
    uint8_t* sel = (uint8_t*)calloc((size_t)n_nodes, 1);
    const int col = has_fg ? 1 : 0;                        /* missing FG -> column 0, missing BG -> column 1 */
    for (int i = 0; i < n_nodes; ++i) {
        int rank = 0;
        const float vi = prior[3 * i + col];
        for (int j = 0; j < n_nodes; ++j) {
            const float vj = prior[3 * j + col];
            if (vj > vi || (vj == vi && j > i)) ++rank;
        }
        sel[i] = rank < n_seed;
    }
    const uint8_t lab = has_fg ? 2 : 3;
    for (size_t p = 0; p < P; ++p) if (seg[p] < n_nodes && sel[seg[p]]) trimap[p] = lab;
    free(sel);
}
