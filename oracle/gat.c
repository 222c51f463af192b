/*
 * oracle/gat.c — CPU restatement of GATTrimapNet.forward, eval mode (reference model.py:323-414).
 * TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Follows model.py: InputNorm :191-213, input_proj :342-346, the loop :391-399 (GATv2Conv -> LayerNorm -> GELU ->
 * EdgeInjectionLayer :142-162), skip :388,401, GlobalContextModule :165-188 with _graph_softmax :90-108, head :373-378.
 * GATv2Conv is torch_geometric's (2.x), restated from its documented semantics — the library is absent, parity with it is
 * UNPINNED; tests/torch_ref.py holds an independent PyTorch restatement this file is checked against (1e-4):
 *   x_l = lin_l(x), x_r = lin_r(x) (with bias); self loops are added with fill_value="mean" (the loop's edge attribute is
 *   the mean of the node's incoming edge attributes); for an edge j -> i and head h
 *     m = leaky_relu(x_r[i] + x_l[j] + lin_edge(e_ij), 0.2);  a = sum_c m[h,c] att[h,c];  alpha = softmax over edges into i
 *     out_i[h] = sum_j alpha_ij x_l[j][h];  heads concatenated; + bias.
 * Float sums follow THE ORDER OF THE MI355X KERNELS (csrc/ggc_resgcn.hip, "GATTrimapNet"), with the shared exp / sigmoid /
 * GELU sequences of include/ggc_fmath.h, so that both produce the same bits (see oracle/resgcn.c's header).
 *
 * Parameter order (P = ggo_gat_n_params(n)): 0-3 in_norm.norm.{weight,bias,running_mean,running_var}; 4,5 input_proj.0.{weight,
 * bias}; 6,7 input_proj.1.{weight,bias}; per layer l at 8 + 13 l: convs.l.{att, lin_l.weight, lin_l.bias, lin_r.weight,
 * lin_r.bias, lin_edge.weight, bias}, lns.l.{weight,bias}, edge_gates.l.proj.{0.weight, 0.bias, 2.weight, 2.bias}; then
 * skip_proj.weight, ctx.attn.{weight,bias}, ctx.compress.{weight,bias}, ctx.expand.{weight,bias}, head.0.{weight,bias},
 * head.3.{weight,bias}.
 */
#include "ggc_oracle.h"
#include "../include/ggc_fmath.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define IN_CH 19
#define EDGE_CH 5
#define N_CLS 3
#define MAX_HEADS 8

int ggo_gat_n_params(int n_layers) { return 8 + 13 * n_layers + 11; }

static void linear_row(const float* x, const float* W, const float* b, int in, int out, float* y) {
    for (int o = 0; o < out; ++o) {
        float acc = 0.0f;
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) acc += x[k] * w[k];
        y[o] = b ? acc + b[o] : acc;
    }
}
static float butterfly(float* v, int width) {          /* v[l] += v[l ^ o], o = width/2 .. 1 */
    float t[64];
    for (int o = width / 2; o > 0; o >>= 1) {
        for (int l = 0; l < width; ++l) t[l] = v[l] + v[l ^ o];
        memcpy(v, t, (size_t)width * sizeof(float));
    }
    return v[0];
}
static void ln_stats_wave(const float* x, int D, float* mean_out, float* rstd_out) {     /* lane l holds columns l, l + 64 */
    float v[64];
    for (int l = 0; l < 64; ++l) { float s1 = 0.0f; for (int c = l; c < D; c += 64) s1 += x[c]; v[l] = s1; }
    const float mean = butterfly(v, 64) / (float)D;
    for (int l = 0; l < 64; ++l) { float s2 = 0.0f; for (int c = l; c < D; c += 64) { const float d = x[c] - mean; s2 += d * d; } v[l] = s2; }
    *mean_out = mean;
    *rstd_out = 1.0f / sqrtf(butterfly(v, 64) / (float)D + 1e-5f);
}
static void mfma_row(const float* a, const float* W, int D, float* acc) {                 /* see oracle/resgcn.c */
    const int KH = D / 2;
    for (int o = 0; o < D; ++o) {
        const float* w = W + (size_t)o * D;
        float c = acc[o];
        for (int s_ = 0; s_ < KH; ++s_) { c = fmaf(a[s_], w[s_], c); c = fmaf(a[KH + s_], w[KH + s_], c); }
        acc[o] = c;
    }
}

/* attention logits of one edge for all heads: lg[h].  A head's C = D / heads channels sit on C consecutive lanes (one
 * butterfly); with ONE head at D = 128 the head spans both registers of a lane: low 64 channels + high 64 channels. */
static void edge_logits(const float* attr, const float* xri, const float* xlj, const float* We /*[D,5]*/, const float* att, int D,
                        int heads, float* lg) {
    const int C = D / heads;
    for (int h = 0; h < heads; ++h) {
        float v[128];
        for (int cc = 0; cc < C; ++cc) {
            const int c = h * C + cc;
            float ev = 0.0f;
            for (int k = 0; k < EDGE_CH; ++k) ev += attr[k] * We[(size_t)c * EDGE_CH + k];
            float m = (xri[c] + xlj[c]) + ev;
            m = m > 0.0f ? m : 0.2f * m;
            v[cc] = m * att[c];
        }
        lg[h] = C <= 64 ? butterfly(v, C) : butterfly(v, 64) + butterfly(v + 64, 64);
    }
}

int ggo_gat_forward(const float* const* P, int D, int heads, int n_layers, int N, int E, const float* x, const int64_t* edge_index,
                    const float* edge_attr, const int64_t* batch, int n_graphs, float* logits, float* probs) {
    if (!(D == 32 || D == 64 || D == 128)) return -2;
    if (!(heads == 1 || heads == 2 || heads == 4 || heads == 8)) return -2;
    const int HEADS = heads;
    const int C = D / HEADS, Dh = D / 2;
    const int64_t* src = edge_index;
    const int64_t* dst = edge_index + E;
    if (!batch) n_graphs = 1;
    const size_t ND = (size_t)N * D;
    float* h = (float*)malloc(ND * sizeof(float));
    float* h2 = (float*)malloc(ND * sizeof(float));
    float* skip = (float*)calloc(ND, sizeof(float));
    float* xl = (float*)malloc(ND * sizeof(float));
    float* xr = (float*)malloc(ND * sizeof(float));
    float* act = (float*)malloc(ND * sizeof(float));
    float* gsum = (float*)malloc(ND * sizeof(float));
    float* buf = (float*)malloc((size_t)(6 * D + 64) * sizeof(float));
    /* incoming edges of every node in edge order (the CSR the kernels build) */
    int* row_ptr = (int*)calloc((size_t)N + 2, sizeof(int));
    int* eids = (int*)malloc((size_t)(E > 0 ? E : 1) * sizeof(int));
    if (!h || !h2 || !skip || !xl || !xr || !act || !gsum || !buf || !row_ptr || !eids) return -1;
    for (int e = 0; e < E; ++e) row_ptr[dst[e] + 2]++;
    for (int i = 0; i < N; ++i) row_ptr[i + 2] += row_ptr[i + 1];
    for (int e = 0; e < E; ++e) eids[row_ptr[dst[e] + 1]++] = e;      /* row_ptr[i+1] ends at the end of row i: rows are row_ptr[i] .. row_ptr[i+1] */

    /* ---- in_norm + input_proj (Linear, LayerNorm, GELU)   [k_gat_input] */
    for (int i = 0; i < N; ++i) {
        float xn[IN_CH];
        for (int k = 0; k < IN_CH; ++k) xn[k] = (x[(size_t)i * IN_CH + k] - P[2][k]) / sqrtf(P[3][k] + 1e-5f) * P[0][k] + P[1][k];
        linear_row(xn, P[4], P[5], IN_CH, D, buf);
        float mean, rstd;
        ln_stats_wave(buf, D, &mean, &rstd);
        for (int k = 0; k < D; ++k) h[(size_t)i * D + k] = ggc_geluf((buf[k] - mean) * rstd * P[6][k] + P[7][k]);
    }
    const float* const* T = P + 8 + 13 * n_layers;          /* tail parameters */
    for (int i = 0; i < N; ++i) mfma_row(h + (size_t)i * D, T[0], D, skip + (size_t)i * D);      /* skip_proj (no bias) */

    for (int l = 0; l < n_layers; ++l) {
        const float* const* L = P + 8 + 13 * l;
        const float *att = L[0], *Wl = L[1], *bl = L[2], *Wr = L[3], *br = L[4], *We = L[5], *bias = L[6], *lnw = L[7], *lnb = L[8];
        for (int i = 0; i < N; ++i) {
            float* a = xl + (size_t)i * D; float* b = xr + (size_t)i * D;
            for (int k = 0; k < D; ++k) { a[k] = 0.0f; b[k] = 0.0f; }
            mfma_row(h + (size_t)i * D, Wl, D, a);
            mfma_row(h + (size_t)i * D, Wr, D, b);
        }
        /* ---- GATv2 attention + bias + LayerNorm + GELU   [k_gat_attn] */
        for (int i = 0; i < N; ++i) {
            const int beg = row_ptr[i], end = row_ptr[i + 1], cnt = end - beg;
            float* xri = buf; float* xli = buf + D; float* xlj = buf + 2 * D; float* acc = buf + 3 * D; float* o = buf + 4 * D;
            for (int k = 0; k < D; ++k) { xri[k] = xr[(size_t)i * D + k] + br[k]; xli[k] = xl[(size_t)i * D + k] + bl[k]; }
            float am[EDGE_CH] = {0, 0, 0, 0, 0};
            for (int p = beg; p < end; ++p) for (int k = 0; k < EDGE_CH; ++k) am[k] += edge_attr[(size_t)eids[p] * EDGE_CH + k];
            const float cf = (float)(cnt > 0 ? cnt : 1);
            for (int k = 0; k < EDGE_CH; ++k) am[k] = am[k] / cf;
            float mx[MAX_HEADS], lgs[MAX_HEADS], lg[MAX_HEADS], ssum[MAX_HEADS];
            edge_logits(am, xri, xli, We, att, D, heads, lgs);
            for (int hh = 0; hh < HEADS; ++hh) mx[hh] = lgs[hh];
            for (int p = beg; p < end; ++p) {
                const int64_t j = src[eids[p]];
                for (int k = 0; k < D; ++k) xlj[k] = xl[(size_t)j * D + k] + bl[k];
                edge_logits(edge_attr + (size_t)eids[p] * EDGE_CH, xri, xlj, We, att, D, heads, lg);
                for (int hh = 0; hh < HEADS; ++hh) mx[hh] = fmaxf(mx[hh], lg[hh]);
            }
            for (int hh = 0; hh < HEADS; ++hh) ssum[hh] = 0.0f;
            for (int k = 0; k < D; ++k) acc[k] = 0.0f;
            for (int p = beg; p < end; ++p) {
                const int64_t j = src[eids[p]];
                for (int k = 0; k < D; ++k) xlj[k] = xl[(size_t)j * D + k] + bl[k];
                edge_logits(edge_attr + (size_t)eids[p] * EDGE_CH, xri, xlj, We, att, D, heads, lg);
                for (int hh = 0; hh < HEADS; ++hh) {
                    const float e = ggc_expf(lg[hh] - mx[hh]);
                    ssum[hh] += e;
                    for (int cc = 0; cc < C; ++cc) acc[hh * C + cc] += e * xlj[hh * C + cc];
                }
            }
            for (int hh = 0; hh < HEADS; ++hh) {           /* the self loop last */
                const float e = ggc_expf(lgs[hh] - mx[hh]);
                ssum[hh] += e;
                for (int cc = 0; cc < C; ++cc) acc[hh * C + cc] += e * xli[hh * C + cc];
            }
            for (int k = 0; k < D; ++k) o[k] = acc[k] / (ssum[k / C] + 1e-16f) + bias[k];
            float mean, rstd;
            ln_stats_wave(o, D, &mean, &rstd);
            for (int k = 0; k < D; ++k) act[(size_t)i * D + k] = ggc_geluf((o[k] - mean) * rstd * lnw[k] + lnb[k]);
        }
        /* ---- EdgeInjectionLayer: act * scatter_mean(sigmoid(W2 relu(W1 e + b1) + b2))   [k_gn_edge_gate, multiply-only] */
        memset(gsum, 0, ND * sizeof(float));
        {
            float* e1 = buf; float* e2 = buf + D;
            for (int e = 0; e < E; ++e) {                   /* edge order == CSR order per destination */
                linear_row(edge_attr + (size_t)e * EDGE_CH, L[9], L[10], EDGE_CH, D, e1);
                for (int k = 0; k < D; ++k) { e1[k] = e1[k] > 0.0f ? e1[k] : 0.0f; e2[k] = 0.0f; }
                mfma_row(e1, L[11], D, e2);
                float* g = gsum + (size_t)dst[e] * D;
                for (int k = 0; k < D; ++k) g[k] += ggc_sigmoid_nr(e2[k] + L[12][k]);
            }
        }
        for (int i = 0; i < N; ++i) {
            const int cnt = row_ptr[i + 1] - row_ptr[i];
            const float cf = (float)(cnt > 1 ? cnt : 1);
            for (int k = 0; k < D; ++k) h2[(size_t)i * D + k] = act[(size_t)i * D + k] * (gsum[(size_t)i * D + k] / cf);
        }
        float* t = h; h = h2; h2 = t;
    }

    /* ---- h + skip, GlobalContextModule   [k_gat_score + k_graph_ctx] */
    float* hs = h2;
    float* score = (float*)malloc((size_t)N * sizeof(float));
    float* gs = (float*)malloc((size_t)n_graphs * D * sizeof(float));
    for (int i = 0; i < N; ++i) {
        float v[64];
        for (int k = 0; k < D; ++k) hs[(size_t)i * D + k] = h[(size_t)i * D + k] + skip[(size_t)i * D + k];
        for (int l = 0; l < 64; ++l) { float d = 0.0f; for (int c = l; c < D; c += 64) d += hs[(size_t)i * D + c] * T[1][c]; v[l] = d; }
        score[i] = butterfly(v, 64) + T[2][0];
    }
    {
        const int NG = 256 / D;
        int beg = 0;
        for (int q = 0; q < n_graphs; ++q) {
            int end = beg;
            if (batch) { while (end < N && (int)batch[end] == q) ++end; } else end = N;
            float peak = -INFINITY;
            for (int i = beg; i < end; ++i) if (score[i] > peak) peak = score[i];
            float red[256];
            for (int t = 0; t < 256; ++t) { float sacc = 0.0f; for (int i = beg + t; i < end; i += 256) sacc += ggc_expf(score[i] - peak); red[t] = sacc; }
            for (int o = 128; o > 0; o >>= 1) for (int t = 0; t < o; ++t) red[t] += red[t + o];
            const float tot = red[0] + 1e-12f;
            float* g = buf;
            for (int k = 0; k < D; ++k) {
                float v = 0.0f;
                for (int grp = 0; grp < NG; ++grp) {
                    float acc = 0.0f;
                    for (int i = beg + grp; i < end; i += NG) acc += (ggc_expf(score[i] - peak) / tot) * hs[(size_t)i * D + k];
                    v += acc;
                }
                g[k] = v;
            }
            float* c = buf + D; float* ex = buf + 2 * D;
            linear_row(g, T[3], T[4], D, Dh, c);
            for (int k = 0; k < Dh; ++k) c[k] = c[k] > 0.0f ? c[k] : 0.0f;
            linear_row(c, T[5], NULL, Dh, D, ex);
            for (int k = 0; k < D; ++k) gs[(size_t)q * D + k] = ggc_sigmoidf(ex[k] + T[6][k]);
            beg = end;
        }
    }
    if (getenv("GGO_GAT_DUMP")) {                           /* diagnostics: intermediates of the LAST layer for stage-by-stage comparison */
        FILE* f = fopen(getenv("GGO_GAT_DUMP"), "wb");
        if (f) {
            fwrite(skip, sizeof(float), ND, f); fwrite(xl, sizeof(float), ND, f); fwrite(xr, sizeof(float), ND, f);
            fwrite(act, sizeof(float), ND, f); fwrite(h, sizeof(float), ND, f); fwrite(hs, sizeof(float), ND, f);
            fwrite(score, sizeof(float), (size_t)N, f); fwrite(gs, sizeof(float), (size_t)n_graphs * D, f);
            fclose(f);
        }
    }
    /* ---- head: Linear, GELU, Linear (model.py:373-378) and softmax   [k_gemm mode 4] */
    for (int i = 0; i < N; ++i) {
        const int q = batch ? (int)batch[i] : 0;
        float* a = buf; float* f = buf + D;
        for (int k = 0; k < D; ++k) { a[k] = hs[(size_t)i * D + k] * gs[(size_t)q * D + k]; f[k] = 0.0f; }
        mfma_row(a, T[7], D, f);
        for (int k = 0; k < D; ++k) f[k] = ggc_geluf(f[k] + T[8][k]);
        float lg[N_CLS];
        for (int c = 0; c < N_CLS; ++c) {
            float v[32];
            const float* hw = T[9] + (size_t)c * D;
            for (int l = 0; l < 32; ++l) { float p = 0.0f; for (int k = l; k < D; k += 32) p += f[k] * hw[k]; v[l] = p; }
            lg[c] = butterfly(v, 32) + T[10][c];
        }
        if (logits) for (int c = 0; c < N_CLS; ++c) logits[(size_t)i * N_CLS + c] = lg[c];
        if (probs) {
            const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
            const float e0 = ggc_expf(lg[0] - mx), e1 = ggc_expf(lg[1] - mx), e2 = ggc_expf(lg[2] - mx);
            const float ssum = (e0 + e1) + e2;
            probs[(size_t)i * N_CLS + 0] = e0 / ssum; probs[(size_t)i * N_CLS + 1] = e1 / ssum; probs[(size_t)i * N_CLS + 2] = e2 / ssum;
        }
    }
    free(h); free(h2); free(skip); free(xl); free(xr); free(act); free(gsum); free(buf); free(row_ptr); free(eids); free(score); free(gs);
    return 0;
}
