/*
 * oracle/gcnnet.c — CPU restatement of GCNTrimapNet.forward, eval mode.
 * TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Follows reference src/gcn_grabcut/model.py: _scatter_mean :69-74, EdgeInjectionLayer :142-162, InputNorm :191-213,
 * ResGCNBlock :216-232, GCNTrimapNet :239-316, with the PyG GCNConv semantics of SURVEY.md Appendix A.3 (shared with
 * oracle/resgcn.c through ggo_gcn_aggregate).  Dropout is the identity in eval mode; `skip` is the identity because
 * every block keeps the width.  BatchNorm1d (eval): (x - mean) / sqrt(var + 1e-5) * weight + bias.
 * All arithmetic is float32, one rounding per operation, sums in index order.
 *
 * Parameter order of `params` (P = ggo_gcnnet_n_params(n)):
 *   0-3   in_norm.norm.{weight,bias,running_mean,running_var}[19]
 *   4,5   input_proj.0.{weight[D,19],bias[D]}    6-9 input_proj.1.{weight,bias,running_mean,running_var}[D]
 *   per block i (10 + 10 i ...):  conv.bias[D]  conv.lin.weight[D,D]  bn.{weight,bias,running_mean,running_var}[D]
 *                                 edge_inject.proj.0.{weight[D,5],bias[D]}  edge_inject.proj.2.{weight[D,D],bias[D]}
 *   then  head.0.{weight[D,D(n+1)],bias[D]}  head.1.{weight,bias,running_mean,running_var}[D]
 *         head.4.{weight[D/2,D],bias[D/2]}   head.6.{weight[3,D/2],bias[3]}
 */
#include "ggc_oracle.h"
#include "../include/ggc_fmath.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IN_CH 19
#define EDGE_CH 5
#define N_CLS 3

int ggo_gcnnet_n_params(int n_layers) { return 10 + 10 * n_layers + 10; }

static float sigmoid_f(float x) { return ggc_sigmoidf(x); }   /* include/ggc_fmath.h: the sequence the kernels use */
static float bn_f(float x, const float* w, const float* b, const float* m, const float* v, int k) {
    return (x - m[k]) / sqrtf(v[k] + 1e-5f) * w[k] + b[k];
}
static void linear_row(const float* x, const float* W, const float* b, int in, int out, float* y) {
    for (int o = 0; o < out; ++o) {
        float acc = 0.0f;
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) acc += x[k] * w[k];
        y[o] = b ? acc + b[o] : acc;
    }
}

/* EdgeInjectionLayer gates (model.py:159-162): per destination, the SUM over its incoming edges of
 * sigmoid(W2 relu(W1 a + b1) + b2) and the edge count (the caller divides: _scatter_mean :69-74).
 * Q[0..3] = proj.0.weight[D,5] proj.0.bias proj.2.weight[D,D] proj.2.bias; buf: >= 2 D floats. */
static void edge_injection_gates(int N, int E, int D, const float* const* Q, const float* edge_attr, const int64_t* dst,
                                 float* gates, float* cnt, float* buf) {
    memset(gates, 0, (size_t)N * D * sizeof(float));
    memset(cnt, 0, (size_t)N * sizeof(float));
    float* e1 = buf; float* e2 = buf + D;
    for (int e = 0; e < E; ++e) {
        linear_row(edge_attr + (size_t)e * EDGE_CH, Q[0], Q[1], EDGE_CH, D, e1);
        for (int k = 0; k < D; ++k) e1[k] = e1[k] > 0.0f ? e1[k] : 0.0f;
        linear_row(e1, Q[2], Q[3], D, D, e2);
        float* gr = gates + (size_t)dst[e] * D;
        for (int k = 0; k < D; ++k) gr[k] += ggc_sigmoid_nr(e2[k]);        /* the kernel's form (include/ggc_fmath.h) */
        cnt[dst[e]] += 1.0f;
    }
}

/* EdgeInjectionLayer.forward on its own (model.py:157-162), for the fixtures recorded from the reference's module */
int ggo_edge_injection(const float* const* Q, int D, int N, int E, const float* edge_attr, const int64_t* edge_index,
                       const float* node_updates, float* out) {
    float* gates = (float*)malloc((size_t)N * D * sizeof(float));
    float* cnt = (float*)malloc((size_t)N * sizeof(float));
    float* buf = (float*)malloc((size_t)(2 * D + 64) * sizeof(float));
    if (!gates || !cnt || !buf) return -1;
    edge_injection_gates(N, E, D, Q, edge_attr, edge_index + E, gates, cnt, buf);
    for (int i = 0; i < N; ++i) {
        const float c = cnt[i] > 1.0f ? cnt[i] : 1.0f;
        for (int k = 0; k < D; ++k) out[(size_t)i * D + k] = node_updates[(size_t)i * D + k] * (gates[(size_t)i * D + k] / c);
    }
    free(gates); free(cnt); free(buf);
    return 0;
}

int ggo_gcnnet_forward(const float* const* P, int D, int n_layers, int N, int E, const float* x, const int64_t* edge_index,
                       const float* edge_attr, float* logits, float* probs) {
    const int Dh = D / 2, n_states = n_layers + 1;
    const int64_t* dst = edge_index + E;
    const size_t ND = (size_t)N * D;
    float* states = (float*)malloc(ND * n_states * sizeof(float));
    float* xw = (float*)malloc(ND * sizeof(float));
    float* conv = (float*)malloc(ND * sizeof(float));
    float* gates = (float*)malloc(ND * sizeof(float));
    float* cnt = (float*)malloc((size_t)N * sizeof(float));
    float* buf = (float*)malloc((size_t)(3 * D + (size_t)D * n_states + 64) * sizeof(float));
    if (!states || !xw || !conv || !gates || !cnt || !buf) return -1;

    /* ---- in_norm + input_proj (Linear, BatchNorm1d, ReLU) — model.py:270-276, :297 */
    for (int i = 0; i < N; ++i) {
        float xn[IN_CH];
        for (int k = 0; k < IN_CH; ++k) xn[k] = bn_f(x[(size_t)i * IN_CH + k], P[0], P[1], P[2], P[3], k);
        linear_row(xn, P[4], P[5], IN_CH, D, buf);
        for (int k = 0; k < D; ++k) {
            const float v = bn_f(buf[k], P[6], P[7], P[8], P[9], k);
            states[(size_t)i * D + k] = v > 0.0f ? v : 0.0f;
        }
    }
    /* ---- residual blocks — ResGCNBlock.forward model.py:225-232 */
    for (int l = 0; l < n_layers; ++l) {
        const float* const* B = P + 10 + 10 * l;
        const float* h = states + ND * l;
        float* out = states + ND * (l + 1);
        for (int i = 0; i < N; ++i) linear_row(h + (size_t)i * D, B[1], NULL, D, D, xw + (size_t)i * D);
        ggo_gcn_aggregate(N, E, D, xw, edge_index, B[0], NULL, NULL, conv);
        edge_injection_gates(N, E, D, B + 6, edge_attr, dst, gates, cnt, buf);
        for (int i = 0; i < N; ++i) {
            const float c = cnt[i] > 1.0f ? cnt[i] : 1.0f;
            for (int k = 0; k < D; ++k) {
                float v = bn_f(conv[(size_t)i * D + k], B[2], B[3], B[4], B[5], k);
                v = v > 0.0f ? v : 0.0f;
                v = v + h[(size_t)i * D + k];
                out[(size_t)i * D + k] = v * (gates[(size_t)i * D + k] / c);
            }
        }
    }
    /* ---- head on the concatenation of all block outputs — model.py:304, :282-290 */
    {
        const float* const* H = P + 10 + 10 * n_layers;
        float* cat = buf + 3 * D;
        float* a = buf; float* b2 = buf + D;
        for (int i = 0; i < N; ++i) {
            for (int s = 0; s < n_states; ++s) memcpy(cat + (size_t)s * D, states + ND * s + (size_t)i * D, (size_t)D * sizeof(float));
            linear_row(cat, H[0], H[1], D * n_states, D, a);
            for (int k = 0; k < D; ++k) { const float v = bn_f(a[k], H[2], H[3], H[4], H[5], k); a[k] = v > 0.0f ? v : 0.0f; }
            linear_row(a, H[6], H[7], D, Dh, b2);
            for (int k = 0; k < Dh; ++k) b2[k] = b2[k] > 0.0f ? b2[k] : 0.0f;
            float lg[N_CLS];
            linear_row(b2, H[8], H[9], Dh, N_CLS, lg);
            if (logits) for (int c = 0; c < N_CLS; ++c) logits[(size_t)i * N_CLS + c] = lg[c];
            if (probs) {
                const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
                const float e0 = ggc_expf(lg[0] - mx), e1v = ggc_expf(lg[1] - mx), e2v = ggc_expf(lg[2] - mx);
                const float s = (e0 + e1v) + e2v;
                probs[(size_t)i * N_CLS + 0] = e0 / s; probs[(size_t)i * N_CLS + 1] = e1v / s; probs[(size_t)i * N_CLS + 2] = e2v / s;
            }
        }
    }
    free(states); free(xw); free(conv); free(gates); free(cnt); free(buf);
    return 0;
}
