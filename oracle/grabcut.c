/*
 * oracle/grabcut.c — GrabCut: GMM colour models + min-cut on the 8-neighbour
 * pixel graph.  TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Follows reference src/gcn_grabcut/grabcut.py: run_with_bbox :81-102,
 * run_with_trimap :104-151 (promotions :127-133, degenerate guard :135-140),
 * refine :153-163, _binary :165-168 — i.e. cv2.grabCut, which is absent here
 * (PARITY UNPINNED) and restated from SURVEY.md Appendix A.4 (OpenCV 4.x
 * imgproc/src/grabcut.cpp): calcBeta, calcNWeights, initGMMs, per iteration
 * assignGMMsComponents / learnGMMs / constructGCGraph / maxFlow /
 * estimateSegmentation; model layout [5 coefs | 15 means | 45 covs].
 *
 * Deliberate, documented differences from OpenCV (DESIGN.md "GrabCut"):
 *  1. initGMMs: OpenCV seeds k-means++ from the process-global theRNG(), which
 *     cannot be reproduced.  Here: seeded k-means++ on the uint8 colours with
 *     exact integer D^2 sampling, 10 assignment steps, exact integer sums.
 *  2. Capacities are quantised to int32 (scale 2^18) after cancelling each
 *     pixel's two t-links against each other and clamping the difference to
 *     +-lambda (no cut changes: lambda = 9*gamma exceeds any node's n-link sum).
 *     Max-flow is then exact integer arithmetic and the min cut is canonical.
 *  3. Labelling: a pixel is foreground iff it cannot reach the sink in the
 *     residual graph (the maximal source set).  OpenCV keeps never-reached
 *     vertices on the source side too, but resolves other free vertices by
 *     search history.
 *  exp/log are the deterministic restatements of mathfn.c.
 * The max-flow itself is a Boykov-Kolmogorov augmenting-path search written
 * for the implicit 8-neighbour grid; its value is pinned against
 * scipy.sparse.csgraph.maximum_flow in tests/test_grabcut_oracle.py.
 */
#include "ggc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GC_BGD 0
#define GC_FGD 1
#define GC_PR_BGD 2
#define GC_PR_FGD 3
#define NCOMP 5
#define CAP_SCALE 262144.0 /* 2^18 */
#define GAMMA 50.0
#define LAMBDA (9.0 * GAMMA)

/* ------------------------------------------------------------------ GMM */
typedef struct {
    double coef[NCOMP], mean[NCOMP][3], cov[NCOMP][9];
    double inv[NCOMP][3][3], det[NCOMP];
} gmm_t;

static void gmm_prepare(gmm_t* g, int ci, double fix) { /* calcInverseCovAndDeterm */
    if (!(g->coef[ci] > 0.0)) return;
    double* c = g->cov[ci];
    double d = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]);
    if (d <= 1e-6 && fix > 0.0) {
        c[0] += fix; c[4] += fix; c[8] += fix;
        d = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]);
    }
    g->det[ci] = d;
    const double id = 1.0 / d;
    g->inv[ci][0][0] = (c[4] * c[8] - c[5] * c[7]) * id;
    g->inv[ci][1][0] = -(c[3] * c[8] - c[5] * c[6]) * id;
    g->inv[ci][2][0] = (c[3] * c[7] - c[4] * c[6]) * id;
    g->inv[ci][0][1] = -(c[1] * c[8] - c[2] * c[7]) * id;
    g->inv[ci][1][1] = (c[0] * c[8] - c[2] * c[6]) * id;
    g->inv[ci][2][1] = -(c[0] * c[7] - c[1] * c[6]) * id;
    g->inv[ci][0][2] = (c[1] * c[5] - c[2] * c[4]) * id;
    g->inv[ci][1][2] = -(c[0] * c[5] - c[2] * c[3]) * id;
    g->inv[ci][2][2] = (c[0] * c[4] - c[1] * c[3]) * id;
}

static void gmm_from_model(gmm_t* g, const double* m) {
    memcpy(g->coef, m, sizeof(g->coef));
    memcpy(g->mean, m + NCOMP, sizeof(g->mean));
    memcpy(g->cov, m + 4 * NCOMP, sizeof(g->cov));
    for (int ci = 0; ci < NCOMP; ++ci) gmm_prepare(g, ci, 0.0);
}

static void gmm_to_model(const gmm_t* g, double* m) {
    memcpy(m, g->coef, sizeof(g->coef));
    memcpy(m + NCOMP, g->mean, sizeof(g->mean));
    memcpy(m + 4 * NCOMP, g->cov, sizeof(g->cov));
}

static double gmm_comp(const gmm_t* g, int ci, const uint8_t* px) {
    if (!(g->coef[ci] > 0.0)) return 0.0;
    const double d0 = (double)px[0] - g->mean[ci][0], d1 = (double)px[1] - g->mean[ci][1], d2 = (double)px[2] - g->mean[ci][2];
    const double mult = d0 * (d0 * g->inv[ci][0][0] + d1 * g->inv[ci][1][0] + d2 * g->inv[ci][2][0])
                      + d1 * (d0 * g->inv[ci][0][1] + d1 * g->inv[ci][1][1] + d2 * g->inv[ci][2][1])
                      + d2 * (d0 * g->inv[ci][0][2] + d1 * g->inv[ci][1][2] + d2 * g->inv[ci][2][2]);
    return 1.0 / sqrt(g->det[ci]) * ggo_exp(-0.5 * mult);
}

static double gmm_total(const gmm_t* g, const uint8_t* px) {
    double r = 0.0;
    for (int ci = 0; ci < NCOMP; ++ci) r += g->coef[ci] * gmm_comp(g, ci, px);
    return r;
}

static int gmm_which(const gmm_t* g, const uint8_t* px) {
    int k = 0;
    double mx = 0.0;
    for (int ci = 0; ci < NCOMP; ++ci) { const double p = gmm_comp(g, ci, px); if (p > mx) { k = ci; mx = p; } }
    return k;
}

/* learning from exact integer sums: n, sum c, sum c c^T per component */
static void gmm_learn(gmm_t* g, const int64_t cnt[NCOMP], const int64_t sum[NCOMP][3], const int64_t prod[NCOMP][9]) {
    int64_t total = 0;
    for (int ci = 0; ci < NCOMP; ++ci) total += cnt[ci];
    for (int ci = 0; ci < NCOMP; ++ci) {
        const int64_t n = cnt[ci];
        if (n == 0) { g->coef[ci] = 0.0; continue; }
        const double inv_n = 1.0 / (double)n;
        g->coef[ci] = (double)n / (double)total;
        for (int i = 0; i < 3; ++i) g->mean[ci][i] = (double)sum[ci][i] * inv_n;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                g->cov[ci][3 * i + j] = (double)prod[ci][3 * i + j] * inv_n - g->mean[ci][i] * g->mean[ci][j];
        gmm_prepare(g, ci, 0.01);
    }
}

/* ------------------------------------------------ seeded k-means++ (initGMMs) */
static uint64_t splitmix(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* samples: pixels p with cls[p] == which, in raster order.  labels[p] for those pixels. */
static void kmeans_init(size_t P, const uint8_t* img, const uint8_t* cls, int which, uint64_t seed, int32_t* labels) {
    size_t n = 0;
    for (size_t p = 0; p < P; ++p) n += cls[p] == which;
    if (n == 0) return;
    const int K = n < NCOMP ? (int)n : NCOMP;
    uint64_t rs = seed * 2 + (uint64_t)which + 1;
    double cen[NCOMP][3];
    int64_t* d2 = (int64_t*)malloc(P * sizeof(int64_t));
    for (int k = 0; k < K; ++k) {
        size_t pick = 0; /* index among the samples */
        if (k == 0) {
            pick = (size_t)(splitmix(&rs) % n);
        } else {
            int64_t total = 0;
            for (size_t p = 0; p < P; ++p) if (cls[p] == which) total += d2[p];
            const uint64_t r = splitmix(&rs);
            if (total == 0) pick = (size_t)(r % n);
            else {
                const int64_t t = (int64_t)(r % (uint64_t)total);
                int64_t run = 0; size_t i = 0; pick = n - 1;
                for (size_t p = 0; p < P; ++p) if (cls[p] == which) { run += d2[p]; if (run > t) { pick = i; break; } ++i; }
            }
        }
        size_t i = 0, pp = 0;
        for (size_t p = 0; p < P; ++p) if (cls[p] == which) { if (i == pick) { pp = p; break; } ++i; }
        for (int c = 0; c < 3; ++c) cen[k][c] = (double)img[3 * pp + c];
        for (size_t p = 0; p < P; ++p) if (cls[p] == which) {
            int64_t d = 0;
            for (int c = 0; c < 3; ++c) { const int64_t t = (int64_t)img[3 * p + c] - (int64_t)img[3 * pp + c]; d += t * t; }
            if (k == 0 || d < d2[p]) d2[p] = d;
        }
    }
    for (int it = 0; it < 10; ++it) {
        int64_t cnt[NCOMP] = {0}, sum[NCOMP][3] = {{0}};
        for (size_t p = 0; p < P; ++p) if (cls[p] == which) {
            int best = 0; double bd = 0.0;
            for (int k = 0; k < K; ++k) {
                const double a = (double)img[3 * p] - cen[k][0], b = (double)img[3 * p + 1] - cen[k][1], c = (double)img[3 * p + 2] - cen[k][2];
                const double d = (a * a + b * b) + c * c;
                if (k == 0 || d < bd) { best = k; bd = d; }
            }
            labels[p] = best;
            cnt[best] += 1;
            for (int c = 0; c < 3; ++c) sum[best][c] += img[3 * p + c];
        }
        if (it < 9)
            for (int k = 0; k < K; ++k)
                if (cnt[k] > 0) for (int c = 0; c < 3; ++c) cen[k][c] = (double)sum[k][c] / (double)cnt[k];
    }
    free(d2);
}

/* ------------------------------------------------------------ max-flow (BK) */
/* directions: 0 left, 1 right, 2 up, 3 down, 4 up-left, 5 down-right, 6 up-right, 7 down-left; rev(d) = d ^ 1 */
static const int DX[8] = {-1, 1, 0, 0, -1, 1, 1, -1};
static const int DY[8] = {0, 0, -1, 1, -1, 1, -1, 1};

#define T_NONE 0
#define T_SRC 1
#define T_SNK 2
#define PAR_NONE (-1)
#define PAR_TERMINAL (-2)
#define PAR_ORPHAN (-3)

typedef struct {
    int H, W, P;
    int32_t* rc;      /* [8][P] residual capacity of arc p -> neighbour(d) */
    int32_t* tw;      /* [P] residual t-link: > 0 from the source, < 0 to the sink */
    uint8_t* tree;
    int8_t* par;      /* direction towards the parent, or PAR_* */
    int32_t *aq, aq_head, aq_tail, aq_cap;   /* active FIFO (circular) */
    uint8_t* in_aq;
    int32_t *oq, oq_n, oq_cap;               /* orphan stack processed FIFO */
    int32_t *ts, *dist, time;                /* adoption-phase timestamps and terminal distances */
} bk_t;

static int nb(const bk_t* g, int p, int d) {
    const int y = p / g->W + DY[d], x = p % g->W + DX[d];
    if (x < 0 || x >= g->W || y < 0 || y >= g->H) return -1;
    return y * g->W + x;
}
static void aq_push(bk_t* g, int p) {
    if (g->in_aq[p]) return;
    g->in_aq[p] = 1;
    g->aq[g->aq_tail] = p; g->aq_tail = (g->aq_tail + 1) % g->aq_cap;
}
/* Distance of q to its terminal along parent links, or -1 if the path hits an orphan.  Nodes verified
 * during the current adoption phase carry ts == time and their distance (the Boykov-Kolmogorov
 * timestamp heuristic), so repeated walks stop early. */
static int origin_dist(bk_t* g, int q) {
    int d = 0, p = q;
    for (;;) {
        if (g->ts[p] == g->time) { d += g->dist[p]; break; }
        const int dir = g->par[p];
        if (dir == PAR_TERMINAL) { g->ts[p] = g->time; g->dist[p] = 1; d += 1; break; }
        if (dir < 0) return -1;
        ++d;
        p = nb(g, p, dir);
    }
    int dd = d;
    for (p = q; g->ts[p] != g->time; p = nb(g, p, g->par[p])) { g->ts[p] = g->time; g->dist[p] = dd--; }
    return d;
}
static void oq_push(bk_t* g, int p) {
    if (g->oq_n == g->oq_cap) { g->oq_cap *= 2; g->oq = (int32_t*)realloc(g->oq, (size_t)g->oq_cap * sizeof(int32_t)); }
    g->oq[g->oq_n++] = p;
}

static int64_t bk_maxflow(bk_t* g) {
    const int P = g->P;
    int64_t flow = 0;
    for (int p = 0; p < P; ++p) {
        g->par[p] = PAR_NONE; g->tree[p] = T_NONE;
        if (g->tw[p] > 0) { g->tree[p] = T_SRC; g->par[p] = PAR_TERMINAL; aq_push(g, p); }
        else if (g->tw[p] < 0) { g->tree[p] = T_SNK; g->par[p] = PAR_TERMINAL; aq_push(g, p); }
    }
    int cur = -1;
    for (;;) {
        /* ---- growth */
        int s_node = -1, t_node = -1, s_dir = -1;
        for (;;) {
            if (cur < 0) {
                if (g->aq_head == g->aq_tail) break;
                cur = g->aq[g->aq_head];
            }
            if (g->tree[cur] == T_NONE) { /* became free while waiting (or during the last adoption) */
                g->in_aq[cur] = 0; g->aq_head = (g->aq_head + 1) % g->aq_cap; cur = -1; continue;
            }
            const int p = cur;
            const int tr = g->tree[p];
            int found = 0;
            for (int d = 0; d < 8 && !found; ++d) {
                const int q = nb(g, p, d);
                if (q < 0) continue;
                const int32_t cap = (tr == T_SRC) ? g->rc[(size_t)d * P + p] : g->rc[(size_t)(d ^ 1) * P + q];
                if (cap <= 0) continue;
                if (g->tree[q] == T_NONE) {
                    g->tree[q] = (uint8_t)tr; g->par[q] = (int8_t)(d ^ 1); aq_push(g, q);
                } else if (g->tree[q] != tr) {
                    if (tr == T_SRC) { s_node = p; t_node = q; s_dir = d; }
                    else { s_node = q; t_node = p; s_dir = d ^ 1; }
                    found = 1;
                }
            }
            if (found) break;
            g->in_aq[p] = 0; g->aq_head = (g->aq_head + 1) % g->aq_cap; cur = -1;
        }
        if (s_node < 0) break;
        /* ---- augmentation along  source ~> s_node -> t_node ~> sink */
        int32_t bott = g->rc[(size_t)s_dir * P + s_node];
        for (int p = s_node;;) {
            const int d = g->par[p];
            if (d == PAR_TERMINAL) { if (g->tw[p] < bott) bott = g->tw[p]; break; }
            const int q = nb(g, p, d);
            const int32_t c = g->rc[(size_t)(d ^ 1) * P + q];   /* arc parent -> p */
            if (c < bott) bott = c;
            p = q;
        }
        for (int p = t_node;;) {
            const int d = g->par[p];
            if (d == PAR_TERMINAL) { if (-g->tw[p] < bott) bott = -g->tw[p]; break; }
            const int32_t c = g->rc[(size_t)d * P + p];         /* arc p -> parent */
            if (c < bott) bott = c;
            p = nb(g, p, d);
        }
        g->rc[(size_t)s_dir * P + s_node] -= bott;
        g->rc[(size_t)(s_dir ^ 1) * P + t_node] += bott;
        for (int p = s_node;;) {
            const int d = g->par[p];
            if (d == PAR_TERMINAL) { g->tw[p] -= bott; if (g->tw[p] == 0) { g->par[p] = PAR_ORPHAN; oq_push(g, p); } break; }
            const int q = nb(g, p, d);
            g->rc[(size_t)(d ^ 1) * P + q] -= bott;
            g->rc[(size_t)d * P + p] += bott;
            if (g->rc[(size_t)(d ^ 1) * P + q] == 0) { g->par[p] = PAR_ORPHAN; oq_push(g, p); }
            p = q;
        }
        for (int p = t_node;;) {
            const int d = g->par[p];
            if (d == PAR_TERMINAL) { g->tw[p] += bott; if (g->tw[p] == 0) { g->par[p] = PAR_ORPHAN; oq_push(g, p); } break; }
            const int q = nb(g, p, d);
            g->rc[(size_t)d * P + p] -= bott;
            g->rc[(size_t)(d ^ 1) * P + q] += bott;
            if (g->rc[(size_t)d * P + p] == 0) { g->par[p] = PAR_ORPHAN; oq_push(g, p); }
            p = q;
        }
        flow += bott;
        /* ---- adoption */
        g->time += 1;
        for (int oi = 0; oi < g->oq_n; ++oi) {
            const int p = g->oq[oi];
            if (g->par[p] != PAR_ORPHAN) continue;
            const int tr = g->tree[p];
            int newpar = PAR_NONE, best = 0x7fffffff;
            for (int d = 0; d < 8; ++d) {
                const int q = nb(g, p, d);
                if (q < 0 || g->tree[q] != tr) continue;
                const int32_t cap = (tr == T_SRC) ? g->rc[(size_t)(d ^ 1) * P + q] : g->rc[(size_t)d * P + p];
                if (cap <= 0) continue;
                const int dq = origin_dist(g, q);
                if (dq >= 0 && dq < best) { best = dq; newpar = d; }   /* closest valid parent */
            }
            if (newpar >= 0) { g->ts[p] = g->time; g->dist[p] = best + 1; }
            if (newpar >= 0) { g->par[p] = (int8_t)newpar; continue; }
            for (int d = 0; d < 8; ++d) {
                const int q = nb(g, p, d);
                if (q < 0 || g->tree[q] != tr) continue;
                const int32_t cap = (tr == T_SRC) ? g->rc[(size_t)(d ^ 1) * P + q] : g->rc[(size_t)d * P + p];
                if (cap > 0) aq_push(g, q);
                if (g->par[q] == (d ^ 1)) { g->par[q] = PAR_ORPHAN; oq_push(g, q); }
            }
            g->tree[p] = T_NONE; g->par[p] = PAR_NONE;
        }
        g->oq_n = 0;
    }
    return flow;
}

int64_t ggo_grid_maxflow(int H, int W, const int32_t* tw, const int32_t* nw, uint8_t* source_side) {
    const int P = H * W;
    bk_t g;
    g.H = H; g.W = W; g.P = P;
    g.rc = (int32_t*)calloc((size_t)8 * P, sizeof(int32_t));
    g.tw = (int32_t*)malloc((size_t)P * sizeof(int32_t));
    g.tree = (uint8_t*)malloc((size_t)P);
    g.par = (int8_t*)malloc((size_t)P);
    g.aq_cap = P + 1; g.aq = (int32_t*)malloc((size_t)g.aq_cap * sizeof(int32_t)); g.aq_head = g.aq_tail = 0;
    g.in_aq = (uint8_t*)calloc((size_t)P, 1);
    g.oq_cap = 1024; g.oq = (int32_t*)malloc((size_t)g.oq_cap * sizeof(int32_t)); g.oq_n = 0;
    g.ts = (int32_t*)calloc((size_t)P, sizeof(int32_t)); g.dist = (int32_t*)calloc((size_t)P, sizeof(int32_t)); g.time = 0;
    memcpy(g.tw, tw, (size_t)P * sizeof(int32_t));
    /* nw planes: 0 left, 1 up-left, 2 up, 3 up-right; each undirected link feeds both arcs */
    static const int plane_dir[4] = {0, 4, 2, 6};
    for (int k = 0; k < 4; ++k)
        for (int p = 0; p < P; ++p) {
            const int d = plane_dir[k];
            const int q = nb(&g, p, d);
            if (q < 0) continue;
            const int32_t w = nw[(size_t)k * P + p];
            g.rc[(size_t)d * P + p] = w;
            g.rc[(size_t)(d ^ 1) * P + q] = w;
        }
    const int64_t flow = bk_maxflow(&g);
    /* canonical labelling: breadth-first search backwards from the sink over residual arcs */
    if (source_side) {
        int32_t* q = (int32_t*)malloc((size_t)P * sizeof(int32_t));
        int qh = 0, qt = 0;
        for (int p = 0; p < P; ++p) { source_side[p] = 1; if (g.tw[p] < 0) { source_side[p] = 0; q[qt++] = p; } }
        while (qh < qt) {
            const int p = q[qh++];
            for (int d = 0; d < 8; ++d) {
                const int u = nb(&g, p, d);
                if (u < 0 || !source_side[u]) continue;
                if (g.rc[(size_t)(d ^ 1) * P + u] > 0) { source_side[u] = 0; q[qt++] = u; }  /* arc u -> p */
            }
        }
        free(q);
    }
    free(g.rc); free(g.tw); free(g.tree); free(g.par); free(g.aq); free(g.in_aq); free(g.oq); free(g.ts); free(g.dist);
    return flow;
}

/* ------------------------------------------------------------ the iteration */
static int32_t quant(double w) { return (int32_t)rint(w * CAP_SCALE); }

int ggo_grabcut(int H, int W, const uint8_t* img, uint8_t* mask, const int32_t* rect,
                double* bgd_model, double* fgd_model, int n_iter, int mode, uint64_t seed,
                uint8_t* binary) {
    const size_t P = (size_t)H * W;
    int rc = 0;
    if (mode == 1) { /* GC_INIT_WITH_RECT: initMaskWithRect */
        int x0 = rect[0], y0 = rect[1], w = rect[2], h = rect[3];
        if (x0 < 0) { w += x0; x0 = 0; }
        if (y0 < 0) { h += y0; y0 = 0; }
        if (x0 + w > W) w = W - x0;
        if (y0 + h > H) h = H - y0;
        memset(mask, GC_BGD, P);
        for (int y = y0; y < y0 + h; ++y) for (int x = x0; x < x0 + w; ++x) mask[(size_t)y * W + x] = GC_PR_FGD;
    }
    for (size_t p = 0; p < P; ++p) if (mask[p] > 3) return -1; /* checkMask */
    if (mode == 0) { /* run_with_trimap promotions (grabcut.py:127-133) */
        int has1 = 0, has0 = 0;
        for (size_t p = 0; p < P; ++p) { has1 |= mask[p] == GC_FGD; has0 |= mask[p] == GC_BGD; }
        if (!has1) for (size_t p = 0; p < P; ++p) if (mask[p] == GC_PR_FGD) mask[p] = GC_FGD;
        if (!has0) for (size_t p = 0; p < P; ++p) if (mask[p] == GC_PR_BGD) mask[p] = GC_BGD;
        has1 = has0 = 0;
        for (size_t p = 0; p < P; ++p) { has1 |= mask[p] == GC_FGD; has0 |= mask[p] == GC_BGD; }
        if (!has1 || !has0) { rc = 1; goto done; } /* degenerate guard (:135-140) */
    }
    {
        gmm_t bg, fg;
        memset(&bg, 0, sizeof(bg)); memset(&fg, 0, sizeof(fg));
        uint8_t* cls = (uint8_t*)malloc(P);
        int32_t* comp = (int32_t*)calloc(P, sizeof(int32_t));
        for (size_t p = 0; p < P; ++p) cls[p] = (mask[p] == GC_BGD || mask[p] == GC_PR_BGD) ? 0 : 1;
        if (mode == 0 || mode == 1) { /* initGMMs */
            size_t nb0 = 0, nf0 = 0;
            for (size_t p = 0; p < P; ++p) { if (cls[p]) ++nf0; else ++nb0; }
            if (nb0 == 0 || nf0 == 0) { free(cls); free(comp); rc = 1; goto done; }
            kmeans_init(P, img, cls, 0, seed, comp);
            kmeans_init(P, img, cls, 1, seed, comp);
            for (int which = 0; which < 2; ++which) {
                int64_t cnt[NCOMP] = {0}, sum[NCOMP][3] = {{0}}, prod[NCOMP][9] = {{0}};
                for (size_t p = 0; p < P; ++p) if (cls[p] == which) {
                    const int ci = comp[p];
                    cnt[ci]++;
                    for (int i = 0; i < 3; ++i) { sum[ci][i] += img[3 * p + i]; for (int j = 0; j < 3; ++j) prod[ci][3 * i + j] += (int64_t)img[3 * p + i] * img[3 * p + j]; }
                }
                gmm_learn(which ? &fg : &bg, cnt, sum, prod);
            }
        } else {
            gmm_from_model(&bg, bgd_model); gmm_from_model(&fg, fgd_model);
        }
        if (n_iter > 0) {
            /* calcBeta: exact integer sum of squared colour differences */
            int64_t bsum = 0;
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x)
                for (int k = 0; k < 4; ++k) {
                    const int d = (k == 0) ? 0 : (k == 1) ? 4 : (k == 2) ? 2 : 6;
                    const int yy = y + DY[d], xx = x + DX[d];
                    if (xx < 0 || xx >= W || yy < 0) continue;
                    for (int c = 0; c < 3; ++c) { const int t = (int)img[3 * ((size_t)y * W + x) + c] - (int)img[3 * ((size_t)yy * W + xx) + c]; bsum += t * t; }
                }
            double beta = 0.0;
            if (bsum > 0) beta = 1.0 / (2.0 * (double)bsum / (double)(4 * (int64_t)W * H - 3 * W - 3 * H + 2));
            /* calcNWeights, quantised */
            int32_t* nw = (int32_t*)calloc(4 * P, sizeof(int32_t));
            const double gdiv = GAMMA / sqrt(2.0);
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x)
                for (int k = 0; k < 4; ++k) {
                    const int d = (k == 0) ? 0 : (k == 1) ? 4 : (k == 2) ? 2 : 6;
                    const int yy = y + DY[d], xx = x + DX[d];
                    if (xx < 0 || xx >= W || yy < 0) continue;
                    int64_t dd = 0;
                    for (int c = 0; c < 3; ++c) { const int t = (int)img[3 * ((size_t)y * W + x) + c] - (int)img[3 * ((size_t)yy * W + xx) + c]; dd += t * t; }
                    const double w = ((k & 1) ? gdiv : GAMMA) * ggo_exp(-beta * (double)dd);
                    nw[(size_t)k * P + (size_t)y * W + x] = quant(w);
                }
            int32_t* tw = (int32_t*)malloc(P * sizeof(int32_t));
            uint8_t* side = (uint8_t*)malloc(P);
            for (int it = 0; it < n_iter; ++it) {
                /* assignGMMsComponents */
                for (size_t p = 0; p < P; ++p) {
                    cls[p] = (mask[p] == GC_BGD || mask[p] == GC_PR_BGD) ? 0 : 1;
                    comp[p] = gmm_which(cls[p] ? &fg : &bg, img + 3 * p);
                }
                /* learnGMMs */
                for (int which = 0; which < 2; ++which) {
                    int64_t cnt[NCOMP] = {0}, sum[NCOMP][3] = {{0}}, prod[NCOMP][9] = {{0}};
                    for (size_t p = 0; p < P; ++p) if (cls[p] == which) {
                        const int ci = comp[p];
                        cnt[ci]++;
                        for (int i = 0; i < 3; ++i) { sum[ci][i] += img[3 * p + i]; for (int j = 0; j < 3; ++j) prod[ci][3 * i + j] += (int64_t)img[3 * p + i] * img[3 * p + j]; }
                    }
                    gmm_learn(which ? &fg : &bg, cnt, sum, prod);
                }
                /* constructGCGraph: t-links (source - sink), clamped to +-lambda */
                for (size_t p = 0; p < P; ++p) {
                    double d;
                    if (mask[p] == GC_BGD) d = -LAMBDA;
                    else if (mask[p] == GC_FGD) d = LAMBDA;
                    else {
                        const double from_src = -ggo_log(gmm_total(&bg, img + 3 * p));
                        const double to_snk = -ggo_log(gmm_total(&fg, img + 3 * p));
                        d = from_src - to_snk;
                        if (d != d) d = 0.0;                   /* inf - inf */
                        if (d > LAMBDA) d = LAMBDA;
                        if (d < -LAMBDA) d = -LAMBDA;
                    }
                    tw[p] = quant(d);
                }
                ggo_grid_maxflow(H, W, tw, nw, side);
                /* estimateSegmentation */
                for (size_t p = 0; p < P; ++p)
                    if (mask[p] == GC_PR_BGD || mask[p] == GC_PR_FGD) mask[p] = side[p] ? GC_PR_FGD : GC_PR_BGD;
            }
            free(nw); free(tw); free(side);
        }
        gmm_to_model(&bg, bgd_model); gmm_to_model(&fg, fgd_model);
        free(cls); free(comp);
    }
done:
    if (binary) for (size_t p = 0; p < P; ++p) binary[p] = (mask[p] == GC_FGD || mask[p] == GC_PR_FGD) ? 1 : 0;
    return rc;
}
