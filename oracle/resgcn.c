/*
 * oracle/resgcn.c — CPU restatement of ResGCNNet.forward, eval mode.
 * TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Follows reference src/gcn_grabcut/model.py:
 *   _scatter_mean :69-74, _graph_softmax :90-108, EdgeContext :111-139,
 *   GlobalContextModule :165-188, InputNorm :191-213, ResGCNNet :421-546,
 * plus the PyTorch-Geometric layer semantics of SURVEY.md Appendix A.3
 * (GCNConv: add self loops after the original edges, symmetric normalisation,
 * scatter-add in edge order, bias last; SAGEConv: mean aggregation, lin_l with
 * bias on the aggregate, lin_r without bias on the root).
 *
 * All arithmetic is float32.  The reference's float sums have no canonical order (PyTorch's CPU kernels vectorise and
 * block them, PyG scatters in edge order), so this restatement fixes one, and it fixes THE ORDER OF THE MI355X KERNELS
 * (csrc/ggc_resgcn.hip) — with exp / sigmoid / GELU as the shared IEEE sequences of include/ggc_fmath.h — so that the two
 * produce the same bits and trimaps and masks can be compared pixel for pixel end to end:
 *   - a Linear on the vector pipe: products added in k order from 0, bias last (one rounding per multiply and per add);
 *   - a D x D Linear on the matrix pipe (GCNConv / SAGEConv / fuse): v_mfma_f32_32x32x2_f32 is a chain of fused
 *     multiply-adds, k from lanes 0..31 before k from lanes 32..63 (tools/micro/mfma_order.hip), and the kernel feeds the
 *     halves k = s and k = D/2 + s together: acc = fma(a[s], w[s], acc); acc = fma(a[D/2 + s], w[D/2 + s], acc), s upward;
 *   - LayerNorm statistics: per-lane partial sums, then the kernel's butterfly (each form is written out where it is used);
 *   - scatter sums in edge order per destination, the self loop last (PyG's order).
 * Against a PyTorch restatement of the reference module the logits stay within 1e-4 (tests/test_resgcn_oracle.py).
 *
 * Parameter order of `params` (P = ggo_resgcn_n_params(n)):
 *   0 in_norm.norm.weight[19]  1 in_norm.norm.bias[19]
 *   2 in_norm.norm.running_mean[19]  3 in_norm.norm.running_var[19]
 *   4 input_proj.0.weight[D,19] 5 input_proj.0.bias[D]
 *   6 input_proj.1.weight[D]    7 input_proj.1.bias[D]
 *   8 prior_booster.0.weight[Q,3]  9 prior_booster.0.bias[Q]     Q = max(D/4, 8)
 *  10 prior_booster.2.weight[D,Q] 11 prior_booster.2.bias[D]
 *  12 edge_ctx.encode.0.weight[C,5] 13 edge_ctx.encode.0.bias[C] C = max(D/2, 8)
 *  14 edge_ctx.encode.2.weight[C,C] 15 edge_ctx.encode.2.bias[C]
 *  16 edge_ctx.to_gate.0.weight[C]  17 edge_ctx.to_gate.0.bias[C]
 *  18 edge_ctx.to_gate.1.weight[D,C] 19 edge_ctx.to_gate.1.bias[D]
 *  20+4i gcn_layers.i.bias[D]  21+4i gcn_layers.i.lin.weight[D,D]
 *  22+4i norms.i.weight[D]     23+4i norms.i.bias[D]
 *  then: sage.lin_l.weight[D,D] sage.lin_l.bias[D] sage.lin_r.weight[D,D]
 *        sage_norm.weight[D] sage_norm.bias[D] jk_logits[n+2]
 *        ctx.attn.weight[1,D] ctx.attn.bias[1] ctx.compress.weight[D/2,D]
 *        ctx.compress.bias[D/2] ctx.expand.weight[D,D/2] ctx.expand.bias[D]
 *        fuse.0.weight[D] fuse.0.bias[D] fuse.1.weight[D,D] fuse.1.bias[D]
 *        head.weight[3,D] head.bias[3]
 */
#include "ggc_oracle.h"
#include "../include/ggc_fmath.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IN_CH 19
#define EDGE_CH 5
#define N_PRIOR 3
#define N_CLS 3

int ggo_resgcn_n_params(int n_layers) { return 20 + 4 * n_layers + 18; }

static float gelu_f(float x) { return ggc_geluf(x); }         /* nn.GELU (exact erf form) to 4e-7: include/ggc_fmath.h */
static float sigmoid_f(float x) { return ggc_sigmoidf(x); }

/* y[o] = sum_k x[k]*W[o,k] + b[o]   (nn.Linear, W row-major [out,in]); products added in k order, bias last */
static void linear_row(const float* x, const float* W, const float* b, int in, int out, float* y) {
    for (int o = 0; o < out; ++o) {
        float acc = 0.0f;
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) acc += x[k] * w[k];
        y[o] = b ? acc + b[o] : acc;
    }
}

/* butterfly sum over `width` lanes (a power of two <= 64): v[l] += v[l ^ o], o = width/2 .. 1; every lane ends equal */
static float butterfly(float* v, int width) {
    float t[64];
    for (int o = width / 2; o > 0; o >>= 1) {
        for (int l = 0; l < width; ++l) t[l] = v[l] + v[l ^ o];
        memcpy(v, t, (size_t)width * sizeof(float));
    }
    return v[0];
}

/* LayerNorm statistics of a wave that owns one row, lane l holding columns l, l + 64, ... (k_input, k_edge_gate) */
static void ln_stats_wave(const float* x, int D, float* mean_out, float* rstd_out) {
    float v[64];
    for (int l = 0; l < 64; ++l) { float s1 = 0.0f; for (int c = l; c < D; c += 64) s1 += x[c]; v[l] = s1; }
    const float mean = butterfly(v, 64) / (float)D;
    for (int l = 0; l < 64; ++l) { float s2 = 0.0f; for (int c = l; c < D; c += 64) { const float d = x[c] - mean; s2 += d * d; } v[l] = s2; }
    *mean_out = mean;
    *rstd_out = 1.0f / sqrtf(butterfly(v, 64) / (float)D + 1e-5f);
}

/* A width that is not a multiple of 32 runs zero-padded to the next one on the MI355X (the MFMA tiling), with the LayerNorm
 * statistics kept on the true width (csrc/ggc_resgcn.hip, k_input).  The sums below therefore split the row at HALF THE PADDED
 * width, as the kernels do; the padded channels are exact zeros and drop out of every sum (x + 0 = x). */
static int padded_width(int D) { return (D + 31) / 32 * 32; }

/* LayerNorm of k_gemm's prologue: each half of the row summed in order, then the two halves added */
static void layernorm_halves(const float* x, const float* w, const float* b, int D, float* y) {
    const int KH = padded_width(D) / 2;
    float s0 = 0.0f, s1 = 0.0f;
    for (int k = 0; k < KH; ++k) { if (k < D) s0 += x[k]; if (KH + k < D) s1 += x[KH + k]; }
    const float mean = (s0 + s1) / (float)D;
    float q0 = 0.0f, q1 = 0.0f;
    for (int k = 0; k < KH; ++k) {
        if (k < D) { const float d0 = x[k] - mean; q0 += d0 * d0; }
        if (KH + k < D) { const float d1 = x[KH + k] - mean; q1 += d1 * d1; }
    }
    const float rstd = 1.0f / sqrtf((q0 + q1) / (float)D + 1e-5f);
    for (int k = 0; k < D; ++k) y[k] = (x[k] - mean) * rstd * w[k] + b[k];
}

/* statistics of k_gemm's epilogue rows: column 32 t + l on lane l (t upward), then a 32-lane butterfly */
static void ln_stats_cols32(const float* x, int D, float* mean_out, float* rstd_out) {
    float v[32];
    for (int l = 0; l < 32; ++l) { float s1 = 0.0f; for (int c = l; c < D; c += 32) s1 += x[c]; v[l] = s1; }
    const float mean = butterfly(v, 32) / (float)D;
    for (int l = 0; l < 32; ++l) { float s2 = 0.0f; for (int c = l; c < D; c += 32) { const float d = x[c] - mean; s2 += d * d; } v[l] = s2; }
    *mean_out = mean;
    *rstd_out = 1.0f / sqrtf(butterfly(v, 32) / (float)D + 1e-5f);
}

/* acc[o] = fma chain over k of a[k] W[o,k] in the matrix pipe's order (see the header), continuing from acc[] */
static void mfma_row(const float* a, const float* W, int D, float* acc) {
    const int KH = padded_width(D) / 2;
    for (int o = 0; o < D; ++o) {
        const float* w = W + (size_t)o * D;
        float c = acc[o];
        for (int s_ = 0; s_ < KH; ++s_) {
            if (s_ < D) c = fmaf(a[s_], w[s_], c);
            if (KH + s_ < D) c = fmaf(a[KH + s_], w[KH + s_], c);
        }
        acc[o] = c;
    }
}

/* PyG GCNConv aggregation on a precomputed xw:
 * deg_i = 1 + indeg_i, dis = deg^-1/2, messages summed in edge order, the
 * self loop last, then + bias.  Optional fused epilogue of model.py:525-527:
 * out = h + gelu(out * gate). */
void ggo_gcn_aggregate(int N, int E, int D, const float* xw, const int64_t* edge_index,
                       const float* bias, const float* gate, const float* h, float* out) {
    const int64_t* src = edge_index;
    const int64_t* dst = edge_index + E;
    float* deg = (float*)calloc((size_t)N, sizeof(float));
    for (int e = 0; e < E; ++e) deg[dst[e]] += 1.0f;
    for (int i = 0; i < N; ++i) deg[i] = 1.0f / sqrtf(deg[i] + 1.0f); /* now dis */
    memset(out, 0, (size_t)N * D * sizeof(float));
    for (int e = 0; e < E; ++e) {
        float nrm = deg[src[e]] * deg[dst[e]];
        const float* xs = xw + (size_t)src[e] * D;
        float* o = out + (size_t)dst[e] * D;
        for (int k = 0; k < D; ++k) o[k] += nrm * xs[k];
    }
    for (int i = 0; i < N; ++i) {
        float nrm = deg[i] * deg[i];
        const float* xs = xw + (size_t)i * D;
        float* o = out + (size_t)i * D;
        for (int k = 0; k < D; ++k) {
            float v = o[k] + nrm * xs[k];
            if (bias) v += bias[k];
            if (gate) v = h[(size_t)i * D + k] + gelu_f(v * gate[(size_t)i * D + k]);
            o[k] = v;
        }
    }
    free(deg);
}

void ggo_gcn_conv(int N, int E, int D, const float* x, const int64_t* edge_index,
                  const float* W, const float* bias, float* out) {
    float* xw = (float*)malloc((size_t)N * D * sizeof(float));
    for (int i = 0; i < N; ++i) linear_row(x + (size_t)i * D, W, NULL, D, D, xw + (size_t)i * D);
    ggo_gcn_aggregate(N, E, D, xw, edge_index, bias, NULL, NULL, out);
    free(xw);
}

/* InputNorm in eval mode (model.py:191-213): BatchNorm1d with the stored statistics, one row */
static void input_norm_row(const float* xi, const float* w, const float* b, const float* mean, const float* var, float* xn) {
    for (int k = 0; k < IN_CH; ++k) xn[k] = (xi[k] - mean[k]) / sqrtf(var[k] + 1e-5f) * w[k] + b[k];
}

/* EdgeContext.forward (model.py:135-139): per-edge MLP, scatter-mean over dst (_scatter_mean :69-74), to_gate.   [k_edge_gate]
 * Q[0..7] = encode.0.weight[C,5] encode.0.bias encode.2.weight[C,C] encode.2.bias to_gate.0.weight[C] to_gate.0.bias
 * to_gate.1.weight[D,C] to_gate.1.bias.  rowbuf: >= 4 max(C, D) floats.
 * The mean over a node's incoming edges commutes with the second (linear) layer: mean_e(W2 e1 + b2) = W2 mean_e(e1) + b2;
 * a node without incoming edges gets the zero vector (_scatter_mean :69-74). */
static void edge_context(int N, int E, int D, int C, const float* const* Q, const float* edge_attr, const int64_t* dst,
                         float* gate, float* rowbuf) {
    float* e1sum = (float*)calloc((size_t)N * C, sizeof(float));
    int* cnt = (int*)calloc((size_t)N, sizeof(int));
    float* e1 = rowbuf;
    for (int e = 0; e < E; ++e) {                      /* edge order == CSR order per destination */
        linear_row(edge_attr + (size_t)e * EDGE_CH, Q[0], Q[1], EDGE_CH, C, e1);
        float* o = e1sum + (size_t)dst[e] * C;
        for (int k = 0; k < C; ++k) o[k] += gelu_f(e1[k]);
        cnt[dst[e]] += 1;
    }
    float* m = rowbuf;
    float* cv = rowbuf + C;
    float* ln = rowbuf + 2 * C;
    float* gt = rowbuf + 3 * C;
    for (int i = 0; i < N; ++i) {
        const float c = (float)(cnt[i] > 0 ? cnt[i] : 1);
        for (int k = 0; k < C; ++k) m[k] = e1sum[(size_t)i * C + k] / c;
        if (cnt[i] > 0) linear_row(m, Q[2], Q[3], C, C, cv);
        else for (int k = 0; k < C; ++k) cv[k] = 0.0f;
        float mean, rstd;
        {   /* lanes >= C hold zeros in the kernel's wave sums */
            float v[64];
            for (int l = 0; l < 64; ++l) v[l] = l < C ? cv[l] : 0.0f;
            mean = butterfly(v, 64) / (float)C;
            for (int l = 0; l < 64; ++l) { const float d = l < C ? cv[l] - mean : 0.0f; v[l] = d * d; }
            rstd = 1.0f / sqrtf(butterfly(v, 64) / (float)C + 1e-5f);
        }
        for (int k = 0; k < C; ++k) ln[k] = (cv[k] - mean) * rstd * Q[4][k] + Q[5][k];
        linear_row(ln, Q[6], NULL, C, D, gt);
        for (int k = 0; k < D; ++k) gate[(size_t)i * D + k] = sigmoid_f(gt[k] + Q[7][k]);
    }
    free(e1sum); free(cnt);
}

/* GlobalContextModule.forward (model.py:176-188) with _graph_softmax (:90-108): out = h * sigmoid(expand(relu(compress(
 * sum_i softmax_i(attn(h)) h_i)))) per graph.   [k_jk's score + k_graph_ctx]
 * Q[0..5] = attn.weight[1,D] attn.bias[1] compress.weight[D/2,D] compress.bias expand.weight[D,D/2] expand.bias.
 * rowbuf: >= 3 D floats.  weights_out (optional, [N]): the per-graph softmax of the attention scores. */
static void global_context(int N, int D, const float* h, const int64_t* batch, int n_graphs, const float* const* Q,
                           float* out, float* rowbuf, float* weights_out) {
    const int Dh = D / 2;
    const float* aw = Q[0];
    const float* ab = Q[1];
    const int Dp = padded_width(D);                                /* the kernels' width: channels >= D are zeros */
    const int LPR = Dp <= 32 ? 8 : (Dp <= 64 ? 16 : 32);          /* lanes per row, 4 consecutive channels each */
    float* score = (float*)malloc((size_t)N * sizeof(float));
    for (int i = 0; i < N; ++i) {
        float v[32];
        const float* hr = h + (size_t)i * D;
        for (int l = 0; l < LPR; ++l) {
            float p[4];
            for (int u = 0; u < 4; ++u) p[u] = 4 * l + u < D ? hr[4 * l + u] * aw[4 * l + u] : 0.0f;
            v[l] = ((p[0] + p[1]) + p[2]) + p[3];
        }
        score[i] = butterfly(v, LPR) + ab[0];
    }
    /* graphs are contiguous node ranges (PyG Batch); one 256-thread block per graph */
    const int NG = 256 / Dp;
    float* gs = (float*)malloc((size_t)n_graphs * D * sizeof(float));
    int beg = 0;
    for (int q = 0; q < n_graphs; ++q) {
        int end = beg;
        if (batch) { while (end < N && (int)batch[end] == q) ++end; } else end = N;
        float peak = -INFINITY;
        for (int i = beg; i < end; ++i) if (score[i] > peak) peak = score[i];
        float red[256];
        for (int t = 0; t < 256; ++t) { float sacc = 0.0f; for (int i = beg + t; i < end; i += 256) sacc += ggc_expf(score[i] - peak); red[t] = sacc; }
        for (int o = 128; o > 0; o >>= 1) for (int t = 0; t < o; ++t) red[t] += red[t + o];
        const float tot = red[0] + 1e-12f;             /* model.py:108 (absorbed in float32 for any non-empty graph) */
        if (weights_out) for (int i = beg; i < end; ++i) weights_out[i] = ggc_expf(score[i] - peak) / tot;
        float* g = rowbuf;                             /* weighted sum: NG strided groups per channel, then added in group order */
        for (int k = 0; k < D; ++k) {
            float v = 0.0f;
            for (int grp = 0; grp < NG; ++grp) {
                float acc = 0.0f;
                for (int i = beg + grp; i < end; i += NG) acc += (ggc_expf(score[i] - peak) / tot) * h[(size_t)i * D + k];
                v += acc;
            }
            g[k] = v;
        }
        float* c = rowbuf + D;
        linear_row(g, Q[2], Q[3], D, Dh, c);
        for (int k = 0; k < Dh; ++k) c[k] = c[k] > 0.0f ? c[k] : 0.0f;
        float* ex = rowbuf + 2 * D;
        linear_row(c, Q[4], NULL, Dh, D, ex);
        for (int k = 0; k < D; ++k) gs[(size_t)q * D + k] = sigmoid_f(ex[k] + Q[5][k]);
        beg = end;
    }
    for (int i = 0; i < N; ++i) {
        int q = batch ? (int)batch[i] : 0;
        for (int k = 0; k < D; ++k) out[(size_t)i * D + k] = h[(size_t)i * D + k] * gs[(size_t)q * D + k];
    }
    free(score); free(gs);
}

/* The three blocks on their own, for the fixtures recorded from the reference's modules (tests/golden/reference_modules.npz) */
void ggo_input_norm(int N, const float* x, const float* w, const float* b, const float* mean, const float* var, float* out) {
    for (int i = 0; i < N; ++i) input_norm_row(x + (size_t)i * IN_CH, w, b, mean, var, out + (size_t)i * IN_CH);
}
int ggo_edge_context(const float* const* Q, int D, int N, int E, const float* edge_attr, const int64_t* edge_index, float* gate) {
    const int C = (D / 2 > 8) ? D / 2 : 8;
    float* rowbuf = (float*)malloc((size_t)(4 * (C > D ? C : D) + 64) * sizeof(float));
    if (!rowbuf) return -1;
    edge_context(N, E, D, C, Q, edge_attr, edge_index + E, gate, rowbuf);
    free(rowbuf);
    return 0;
}
int ggo_global_context(const float* const* Q, int D, int N, const float* h, const int64_t* batch, int n_graphs, float* out, float* weights) {
    float* rowbuf = (float*)malloc((size_t)(4 * D + 64) * sizeof(float));
    if (!rowbuf) return -1;
    if (!batch) n_graphs = 1;
    global_context(N, D, h, batch, n_graphs, Q, out, rowbuf, weights);
    free(rowbuf);
    return 0;
}

int ggo_resgcn_forward(const float* const* P, int D, int n_layers,
                       int N, int E, const float* x, const int64_t* edge_index,
                       const float* edge_attr, const int64_t* batch, int n_graphs,
                       float* logits, float* probs) {
    const int Q = (D / 4 > 8) ? D / 4 : 8;
    const int C = (D / 2 > 8) ? D / 2 : 8;
    const int64_t* src = edge_index;
    const int64_t* dst = edge_index + E;
    if (!batch) n_graphs = 1;
    size_t ND = (size_t)N * D;
    int n_states = n_layers + 2;
    float* states = (float*)malloc(ND * n_states * sizeof(float));
    float* gate = (float*)malloc(ND * sizeof(float));
    float* tmp = (float*)malloc(ND * sizeof(float));
    float* tmp2 = (float*)malloc(ND * sizeof(float));
    float* rowbuf = (float*)malloc((size_t)(4 * D + 64) * sizeof(float));
    if (!states || !gate || !tmp || !tmp2 || !rowbuf) return -1;

    /* ---- in_norm (BatchNorm1d eval, model.py:191-213) + input_proj + prior_booster (:516-518)   [k_input] */
    float* h = states;
    for (int i = 0; i < N; ++i) {
        const float* xi = x + (size_t)i * IN_CH;
        float xn[IN_CH];
        input_norm_row(xi, P[0], P[1], P[2], P[3], xn);
        float* a = rowbuf;
        float* bq = rowbuf + D;
        linear_row(xn, P[4], P[5], IN_CH, D, a);
        float mean, rstd;
        ln_stats_wave(a, D, &mean, &rstd);
        const float* prior = xi + (IN_CH - N_PRIOR);
        linear_row(prior, P[8], P[9], N_PRIOR, Q, bq);
        for (int k = 0; k < Q; ++k) bq[k] = gelu_f(bq[k]);
        float* g = rowbuf + 2 * D;
        linear_row(bq, P[10], NULL, Q, D, g);
        for (int k = 0; k < D; ++k) {
            const float v = gelu_f((a[k] - mean) * rstd * P[6][k] + P[7][k]);
            h[(size_t)i * D + k] = v * (1.0f + sigmoid_f(g[k] + P[11][k]));
        }
    }

    /* ---- EdgeContext (model.py:135-139)   [k_edge_gate] */
    edge_context(N, E, D, C, P + 12, edge_attr, dst, gate, rowbuf);

    /* ---- residual blocks (model.py:523-528)   [k_gemm mode 0 + k_aggregate_graph] */
    for (int l = 0; l < n_layers; ++l) {
        const float* bias = P[20 + 4 * l];
        const float* W = P[21 + 4 * l];
        const float* h_in = states + ND * l;
        float* h_out = states + ND * (l + 1);
        for (int i = 0; i < N; ++i) {
            layernorm_halves(h_in + (size_t)i * D, P[22 + 4 * l], P[23 + 4 * l], D, rowbuf);
            float* o = tmp + (size_t)i * D;
            for (int k = 0; k < D; ++k) o[k] = 0.0f;
            mfma_row(rowbuf, W, D, o);
        }
        ggo_gcn_aggregate(N, E, D, tmp, edge_index, bias, gate, h_in, h_out);
    }

    int b0 = 20 + 4 * n_layers;
    /* ---- SAGEConv + sage_norm + GELU (model.py:530)   [k_aggregate_graph mode 1 + k_gemm mode 1] */
    {
        const float* hl = states + ND * n_layers;
        float* cnt = (float*)calloc((size_t)N, sizeof(float));
        memset(tmp, 0, ND * sizeof(float));
        for (int e = 0; e < E; ++e) {
            const float* xs = hl + (size_t)src[e] * D;
            float* o = tmp + (size_t)dst[e] * D;
            for (int k = 0; k < D; ++k) o[k] += xs[k];
            cnt[dst[e]] += 1.0f;
        }
        float* so = states + ND * (n_layers + 1);
        for (int i = 0; i < N; ++i) {
            float c = cnt[i] < 1.0f ? 1.0f : cnt[i];
            float* m = tmp + (size_t)i * D;
            for (int k = 0; k < D; ++k) m[k] = m[k] / c;
            float* a = rowbuf;
            for (int k = 0; k < D; ++k) a[k] = 0.0f;
            mfma_row(m, P[b0 + 0], D, a);                  /* lin_l on the aggregate ... */
            mfma_row(hl + (size_t)i * D, P[b0 + 2], D, a); /* ... and lin_r on the root continue ONE accumulator */
            for (int k = 0; k < D; ++k) a[k] += P[b0 + 1][k];
            float mean, rstd;
            ln_stats_cols32(a, D, &mean, &rstd);
            for (int k = 0; k < D; ++k) so[(size_t)i * D + k] = gelu_f((a[k] - mean) * rstd * P[b0 + 3][k] + P[b0 + 4][k]);
        }
        free(cnt);
    }

    /* ---- JK fusion (model.py:532-533)   [host softmax + k_jk] */
    {
        const float* jl = P[b0 + 5];
        float w[64];
        float mx = jl[0];
        for (int k = 1; k < n_states; ++k) if (jl[k] > mx) mx = jl[k];
        float s = 0.0f;
        for (int k = 0; k < n_states; ++k) { w[k] = ggc_expf(jl[k] - mx); s += w[k]; }
        for (int k = 0; k < n_states; ++k) w[k] = w[k] / s;
        for (size_t t = 0; t < ND; ++t) {
            float acc = 0.0f;
            for (int k = 0; k < n_states; ++k) acc += states[ND * k + t] * w[k];
            tmp[t] = acc; /* h_jk */
        }
    }

    /* ---- GlobalContextModule (model.py:176-188) with _graph_softmax (:90-108)   [k_jk's score + k_graph_ctx] */
    global_context(N, D, tmp, batch, n_graphs, P + b0 + 6, tmp2, rowbuf, NULL);

    /* ---- fuse + head (model.py:491-497,536) and softmax (:543-546)   [k_gemm mode 2] */
    for (int i = 0; i < N; ++i) {
        float* a = rowbuf;
        float* f = rowbuf + D;
        layernorm_halves(tmp2 + (size_t)i * D, P[b0 + 12], P[b0 + 13], D, a);
        for (int k = 0; k < D; ++k) f[k] = 0.0f;
        mfma_row(a, P[b0 + 14], D, f);
        for (int k = 0; k < D; ++k) f[k] = gelu_f(f[k] + P[b0 + 15][k]);
        float lg[N_CLS];
        for (int c = 0; c < N_CLS; ++c) {                  /* head: column 32 t + l on lane l, then a 32-lane butterfly */
            float v[32];
            const float* hw = P[b0 + 16] + (size_t)c * D;
            for (int l = 0; l < 32; ++l) { float p = 0.0f; for (int k = l; k < D; k += 32) p += f[k] * hw[k]; v[l] = p; }
            lg[c] = butterfly(v, 32) + P[b0 + 17][c];
        }
        if (logits) for (int c = 0; c < N_CLS; ++c) logits[(size_t)i * N_CLS + c] = lg[c];
        if (probs) {
            const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
            const float e0 = ggc_expf(lg[0] - mx), e1 = ggc_expf(lg[1] - mx), e2 = ggc_expf(lg[2] - mx);
            const float ssum = (e0 + e1) + e2;
            probs[(size_t)i * N_CLS + 0] = e0 / ssum; probs[(size_t)i * N_CLS + 1] = e1 / ssum; probs[(size_t)i * N_CLS + 2] = e2 / ssum;
        }
    }
    free(states); free(gate); free(tmp); free(tmp2); free(rowbuf);
    return 0;
}
