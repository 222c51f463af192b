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
 * All arithmetic is float32, one rounding per operation, sums in index order.
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
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IN_CH 19
#define EDGE_CH 5
#define N_PRIOR 3
#define N_CLS 3

int ggo_resgcn_n_params(int n_layers) { return 20 + 4 * n_layers + 18; }

static float gelu_f(float x) {
    /* torch.nn.GELU (exact erf form) */
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
static float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

/* y[o] = sum_k x[k]*W[o,k] + b[o]   (nn.Linear, W row-major [out,in]) */
static void linear_row(const float* x, const float* W, const float* b, int in, int out, float* y) {
    for (int o = 0; o < out; ++o) {
        float acc = 0.0f;
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) acc += x[k] * w[k];
        y[o] = b ? acc + b[o] : acc;
    }
}

/* nn.LayerNorm over the last dim, eps 1e-5, biased variance */
static void layernorm_row(const float* x, const float* w, const float* b, int D, float* y) {
    float mean = 0.0f;
    for (int k = 0; k < D; ++k) mean += x[k];
    mean /= (float)D;
    float var = 0.0f;
    for (int k = 0; k < D; ++k) { float d = x[k] - mean; var += d * d; }
    var /= (float)D;
    float rstd = 1.0f / sqrtf(var + 1e-5f);
    for (int k = 0; k < D; ++k) y[k] = (x[k] - mean) * rstd * w[k] + b[k];
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

int ggo_resgcn_forward(const float* const* P, int D, int n_layers,
                       int N, int E, const float* x, const int64_t* edge_index,
                       const float* edge_attr, const int64_t* batch, int n_graphs,
                       float* logits, float* probs) {
    const int Q = (D / 4 > 8) ? D / 4 : 8;
    const int C = (D / 2 > 8) ? D / 2 : 8;
    const int Dh = D / 2;
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

    /* ---- in_norm (BatchNorm1d eval, model.py:191-213) + input_proj + prior_booster (:516-518) */
    float* h = states;
    for (int i = 0; i < N; ++i) {
        const float* xi = x + (size_t)i * IN_CH;
        float xn[IN_CH];
        for (int k = 0; k < IN_CH; ++k)
            xn[k] = (xi[k] - P[2][k]) / sqrtf(P[3][k] + 1e-5f) * P[0][k] + P[1][k];
        float* a = rowbuf;
        float* bq = rowbuf + D;
        linear_row(xn, P[4], P[5], IN_CH, D, a);
        layernorm_row(a, P[6], P[7], D, a);
        for (int k = 0; k < D; ++k) a[k] = gelu_f(a[k]);
        const float* prior = xi + (IN_CH - N_PRIOR);
        linear_row(prior, P[8], P[9], N_PRIOR, Q, bq);
        for (int k = 0; k < Q; ++k) bq[k] = gelu_f(bq[k]);
        float* g = rowbuf + 2 * D;
        linear_row(bq, P[10], P[11], Q, D, g);
        for (int k = 0; k < D; ++k) h[(size_t)i * D + k] = a[k] * (1.0f + sigmoid_f(g[k]));
    }

    /* ---- EdgeContext (model.py:135-139): per-edge MLP, scatter-mean over dst, to_gate */
    {
        float* ctx = (float*)calloc((size_t)N * C, sizeof(float));
        float* cnt = (float*)calloc((size_t)N, sizeof(float));
        float* e1 = rowbuf;
        float* e2 = rowbuf + C;
        for (int e = 0; e < E; ++e) {
            linear_row(edge_attr + (size_t)e * EDGE_CH, P[12], P[13], EDGE_CH, C, e1);
            for (int k = 0; k < C; ++k) e1[k] = gelu_f(e1[k]);
            linear_row(e1, P[14], P[15], C, C, e2);
            float* o = ctx + (size_t)dst[e] * C;
            for (int k = 0; k < C; ++k) o[k] += e2[k];
            cnt[dst[e]] += 1.0f;
        }
        for (int i = 0; i < N; ++i) {
            float c = cnt[i] < 1.0f ? 1.0f : cnt[i];
            float* o = ctx + (size_t)i * C;
            for (int k = 0; k < C; ++k) o[k] = o[k] / c;
            layernorm_row(o, P[16], P[17], C, e1);
            linear_row(e1, P[18], P[19], C, D, gate + (size_t)i * D);
            for (int k = 0; k < D; ++k) gate[(size_t)i * D + k] = sigmoid_f(gate[(size_t)i * D + k]);
        }
        free(ctx); free(cnt);
    }

    /* ---- residual blocks (model.py:523-528) */
    for (int l = 0; l < n_layers; ++l) {
        const float* bias = P[20 + 4 * l];
        const float* W = P[21 + 4 * l];
        const float* h_in = states + ND * l;
        float* h_out = states + ND * (l + 1);
        for (int i = 0; i < N; ++i) {
            layernorm_row(h_in + (size_t)i * D, P[22 + 4 * l], P[23 + 4 * l], D, rowbuf);
            linear_row(rowbuf, W, NULL, D, D, tmp + (size_t)i * D);
        }
        ggo_gcn_aggregate(N, E, D, tmp, edge_index, bias, gate, h_in, h_out);
    }

    int b0 = 20 + 4 * n_layers;
    /* ---- SAGEConv + sage_norm + GELU (model.py:530) */
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
            float* r = rowbuf + D;
            linear_row(m, P[b0 + 0], P[b0 + 1], D, D, a);
            linear_row(hl + (size_t)i * D, P[b0 + 2], NULL, D, D, r);
            for (int k = 0; k < D; ++k) a[k] = a[k] + r[k];
            layernorm_row(a, P[b0 + 3], P[b0 + 4], D, a);
            for (int k = 0; k < D; ++k) so[(size_t)i * D + k] = gelu_f(a[k]);
        }
        free(cnt);
    }

    /* ---- JK fusion (model.py:532-533) */
    {
        const float* jl = P[b0 + 5];
        float w[64];
        float mx = jl[0];
        for (int k = 1; k < n_states; ++k) if (jl[k] > mx) mx = jl[k];
        float s = 0.0f;
        for (int k = 0; k < n_states; ++k) { w[k] = expf(jl[k] - mx); s += w[k]; }
        for (int k = 0; k < n_states; ++k) w[k] = w[k] / s;
        for (size_t t = 0; t < ND; ++t) {
            float acc = 0.0f;
            for (int k = 0; k < n_states; ++k) acc += states[ND * k + t] * w[k];
            tmp[t] = acc; /* h_jk */
        }
    }

    /* ---- GlobalContextModule (model.py:176-188) with _graph_softmax (:90-108) */
    {
        const float* aw = P[b0 + 6];
        const float* ab = P[b0 + 7];
        float* score = (float*)malloc((size_t)N * sizeof(float));
        for (int i = 0; i < N; ++i) {
            float acc = 0.0f;
            for (int k = 0; k < D; ++k) acc += tmp[(size_t)i * D + k] * aw[k];
            score[i] = acc + ab[0];
        }
        float* peak = (float*)malloc((size_t)n_graphs * sizeof(float));
        float* tot = (float*)calloc((size_t)n_graphs, sizeof(float));
        float* g = (float*)calloc((size_t)n_graphs * D, sizeof(float));
        for (int q = 0; q < n_graphs; ++q) peak[q] = -INFINITY;
        for (int i = 0; i < N; ++i) { int q = batch ? (int)batch[i] : 0; if (score[i] > peak[q]) peak[q] = score[i]; }
        for (int i = 0; i < N; ++i) { int q = batch ? (int)batch[i] : 0; score[i] = expf(score[i] - peak[q]); tot[q] += score[i]; }
        for (int i = 0; i < N; ++i) {
            int q = batch ? (int)batch[i] : 0;
            /* batch=None: torch.softmax (no epsilon); batched: ex / (tot + 1e-12) (model.py:108) */
            float wgt = batch ? score[i] / (tot[q] + 1e-12f) : score[i] / tot[q];
            for (int k = 0; k < D; ++k) g[(size_t)q * D + k] += wgt * tmp[(size_t)i * D + k];
        }
        float* gs = (float*)malloc((size_t)n_graphs * D * sizeof(float));
        for (int q = 0; q < n_graphs; ++q) {
            float* c = rowbuf;
            linear_row(g + (size_t)q * D, P[b0 + 8], P[b0 + 9], D, Dh, c);
            for (int k = 0; k < Dh; ++k) c[k] = c[k] > 0.0f ? c[k] : 0.0f;
            linear_row(c, P[b0 + 10], P[b0 + 11], Dh, D, gs + (size_t)q * D);
            for (int k = 0; k < D; ++k) gs[(size_t)q * D + k] = sigmoid_f(gs[(size_t)q * D + k]);
        }
        for (int i = 0; i < N; ++i) {
            int q = batch ? (int)batch[i] : 0;
            for (int k = 0; k < D; ++k) tmp2[(size_t)i * D + k] = tmp[(size_t)i * D + k] * gs[(size_t)q * D + k];
        }
        free(score); free(peak); free(tot); free(g); free(gs);
    }

    /* ---- fuse + head (model.py:491-497,536) and softmax (:543-546) */
    for (int i = 0; i < N; ++i) {
        float* a = rowbuf;
        float* f = rowbuf + D;
        layernorm_row(tmp2 + (size_t)i * D, P[b0 + 12], P[b0 + 13], D, a);
        linear_row(a, P[b0 + 14], P[b0 + 15], D, D, f);
        for (int k = 0; k < D; ++k) f[k] = gelu_f(f[k]);
        float lg[N_CLS];
        linear_row(f, P[b0 + 16], P[b0 + 17], D, N_CLS, lg);
        if (logits) for (int c = 0; c < N_CLS; ++c) logits[(size_t)i * N_CLS + c] = lg[c];
        if (probs) {
            float mx = lg[0];
            for (int c = 1; c < N_CLS; ++c) if (lg[c] > mx) mx = lg[c];
            float ex[N_CLS], s = 0.0f;
            for (int c = 0; c < N_CLS; ++c) { ex[c] = expf(lg[c] - mx); s += ex[c]; }
            for (int c = 0; c < N_CLS; ++c) probs[(size_t)i * N_CLS + c] = ex[c] / s;
        }
    }
    free(states); free(gate); free(tmp); free(tmp2); free(rowbuf);
    return 0;
}
