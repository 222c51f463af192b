/*
 * ggc_oracle.h — CPU restatement ("oracle") of the GCN-GrabCut hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load libggc_oracle.so.  The product (gcn-grabcut_amd/) never links, imports
 * or falls back to it.
 *
 * Plain C99, single thread, built with -O2 -ffp-contract=off so every
 * floating-point operation is one IEEE-754 rounding (no FMA contraction); the
 * HIP kernels that must agree bit-for-bit on integer outputs (SLIC label map,
 * trimap) are compiled the same way.
 *
 * Parity status (see DESIGN.md "Oracle pinning"):
 *   - SLIC k-means + connectivity: pinned bit-exact against scikit-image
 *     0.18.3's compiled _slic kernels (tests/golden/slic_*.npz).
 *   - Gaussian pre-smoothing: pinned bit-exact against scipy.ndimage.
 *   - find_boundaries, rgb2lab, rgb2hsv: pinned against scikit-image 0.18.3.
 *   - ResGCNNet forward: pinned against a PyG-free torch-CPU restatement.
 *   - max-flow value: pinned against scipy.sparse.csgraph.maximum_flow.
 *   - cv2.blur / cv2.grabCut / cv2.cvtColor / cv2.connectedComponents /
 *     PyG GCNConv+SAGEConv: PARITY UNPINNED — the libraries are absent here and
 *     the reference holds no golden vectors; restated from SURVEY Appendix A.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference tree).
 */
#ifndef GGC_ORACLE_H
#define GGC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- shared deterministic math (restated independently in the HIP code) ---- */
/* 8-bit HSV (mode 0) / Lab (mode 1) of a BGR image, for GrabCutConfig.color_space (grabcut.py:73-79); see oracle/color.c */
void ggo_convert_color8(size_t n, const uint8_t* bgr, int mode, uint8_t* out);
double ggo_cbrt(double a);        /* a > 0 */
double ggo_pow24(double a);       /* a^2.4, a > 0 */
double ggo_exp(double x);
double ggo_log(double x);

/* ---- G0: GraphBuilder.__init__ (graph_builder.py:142-154) ---- */
void ggo_preprocess(int H, int W, const uint8_t* bgr,
                    float* lab, float* hsv, float* gray, float* grad);

/* ---- G1: skimage slic as called at graph_builder.py:180-187 ---- */
/* individual steps, exposed for pinning against skimage/scipy */
void ggo_slic_rescale_lab(int H, int W, const float* image, int rescale_input, float* out);
/* the float64 instance of the same steps: SuperpixelGraphConfig(use_lab=False) = slic(rgb.astype(float)) (graph_builder.py:177-179) */
void ggo_slic_rescale_lab64(int H, int W, const double* image, int rescale_input, double* out);
void ggo_gaussian_f64(int H, int W, int C, const double* in, double sigma, double* out);
void ggo_slic_kmeans64(int H, int W, const double* image, int K, double* centers, double step, int max_iter, int32_t* labels);
int  ggo_slic_rgb(int H, int W, const uint8_t* bgr, int n_segments, double compactness, double sigma, int32_t* segments);
void ggo_gaussian_f32(int H, int W, int C, const float* in, double sigma, float* out);
int  ggo_gaussian_weights(double sigma, double* w /*[2r+1]*/, int cap); /* returns radius */
int  ggo_slic_grid(int H, int W, int n_segments, int* step_y, int* step_x,
                   int* start_y, int* start_x, int* ny, int* nx);
/* centers: [K,5] = y,x,c0,c1,c2 in/out; labels out; image already scaled by 1/compactness */
void ggo_slic_kmeans(int H, int W, const float* image, int K, float* centers,
                     float step, int max_iter, int32_t* labels);
int  ggo_slic_connectivity(int H, int W, const int32_t* labels, int min_size, int max_size,
                           int32_t* out);
/* whole thing; returns n_nodes */
int  ggo_slic(int H, int W, const float* image, int n_segments, float compactness,
              float sigma, int rescale_input, int32_t* segments);

/* ---- G2-G8: graph_builder.py:190-454 ---- */
void ggo_find_boundaries_inner(int H, int W, const int32_t* seg, uint8_t* out);
/* Two-phase like the product. Returns a handle (opaque), fills n_nodes/n_edges. */
typedef struct ggo_graph ggo_graph;
ggo_graph* ggo_graph_build(int H, int W, const int32_t* segments,
                           const float* lab, const float* hsv, const float* grad,
                           int connectivity, int n_nonlocal, int* n_nodes, int* n_edges);
void ggo_graph_get(const ggo_graph* g, float* node_features /*[N,16]*/, float* prior /*[N,3]*/,
                   float* centroids /*[N,2]*/, float* area_ratio /*[N]*/,
                   int64_t* edge_index /*[2,E]*/, float* edge_attr /*[E,5]*/);
void ggo_graph_free(ggo_graph* g);
void ggo_auto_prior(int H, int W, const int32_t* segments, const float* lab, int n_nodes,
                    float* prior /*[N,3]*/);
void ggo_auto_prior_sigmas(int H, int W, const int32_t* segments, const float* lab, int n_nodes, double centre_sigma,
                           double contrast_sigma, float* prior);

/* ---- M0-M7: ResGCNNet.forward (model.py:508-536), eval mode ---- */
int ggo_resgcn_n_params(int n_layers);
/* network blocks on their own (what the forwards above call), for fixtures recorded from the reference's modules:
 * InputNorm eval (model.py:191-213), EdgeContext (:111-139), GlobalContextModule + _graph_softmax (:90-108,165-188),
 * EdgeInjectionLayer (:142-162).  Q = the module's parameters in state_dict order. */
void ggo_input_norm(int N, const float* x, const float* w, const float* b, const float* mean, const float* var, float* out);
int ggo_edge_context(const float* const* Q, int D, int N, int E, const float* edge_attr, const int64_t* edge_index, float* gate);
int ggo_global_context(const float* const* Q, int D, int N, const float* h, const int64_t* batch, int n_graphs, float* out, float* weights);
int ggo_edge_injection(const float* const* Q, int D, int N, int E, const float* edge_attr, const int64_t* edge_index,
                       const float* node_updates, float* out);
/* GCNTrimapNet.forward, eval mode (reference model.py:239-316); parameter order in oracle/gcnnet.c */
int ggo_gcnnet_n_params(int n_layers);
int ggo_gcnnet_forward(const float* const* params, int D, int n_layers, int N, int E, const float* x,
                       const int64_t* edge_index, const float* edge_attr, float* logits, float* probs);
/* GATTrimapNet (model.py:323-414), heads in {1, 2, 4, 8}, D in {32, 64, 128}; parameter order in gat.c */
int ggo_gat_n_params(int n_layers);
int ggo_gat_forward(const float* const* params, int D, int heads, int n_layers, int N, int E, const float* x, const int64_t* edge_index,
                    const float* edge_attr, const int64_t* batch, int n_graphs, float* logits, float* probs);
int ggo_resgcn_forward(const float* const* params, int D, int n_layers,
                       int N, int E, const float* x, const int64_t* edge_index,
                       const float* edge_attr, const int64_t* batch, int n_graphs,
                       float* logits, float* probs);
/* GCNConv alone (PyG semantics): out = Â (x W^T) + b */
void ggo_gcn_conv(int N, int E, int D, const float* x, const int64_t* edge_index,
                  const float* W, const float* bias, float* out);
/* aggregation only, on a precomputed xw (the graded scatter-gather) */
void ggo_gcn_aggregate(int N, int E, int D, const float* xw, const int64_t* edge_index,
                       const float* bias, const float* gate, const float* h, float* out);

/* ---- P0-P3, S0: pipeline.py:71-186, model.py:623-678 ---- */
void ggo_box_blur(int H, int W, const float* in, int radius, float* out);
void ggo_guided_filter(int H, int W, const float* guide, const float* src, int radius,
                       float eps, float* out);
void ggo_refine_trimap(int H, int W, const float* probs, int n_probs, const int32_t* segments,
                       const uint8_t* bgr, float thr_fg, float thr_bg, int radius, float eps,
                       int edge_aware, uint8_t* trimap);
void ggo_seed_from_prior(int H, int W, const float* prior, int n_nodes, const int32_t* segments,
                         double seed_frac, uint8_t* trimap);

/* ---- C0-C6: grabcut.py:81-168 / cv2.grabCut ---- */
int ggo_grabcut(int H, int W, const uint8_t* image, uint8_t* mask, const int32_t* rect,
                double* bgd_model, double* fgd_model, int n_iter, int mode, uint64_t seed,
                uint8_t* binary);
/* max-flow on the 8-neighbour grid with integer capacities; returns flow value,
 * fills source_side (1 = cannot reach the sink in the residual graph). */
int64_t ggo_grid_maxflow(int H, int W, const int32_t* tw /*[P] source-sink*/,
                         const int32_t* nw /*[4,P] left,upleft,up,upright*/,
                         uint8_t* source_side);

/* ---- K0, O0, R0 ---- */
void ggo_clean_mask(int H, int W, const uint8_t* mask, float min_area_ratio, int keep_largest,
                    uint8_t* out);
void ggo_compose(int H, int W, const uint8_t* bgr, const uint8_t* binary, float alpha,
                 int tb, int tg, int tr, uint8_t* overlay, uint8_t* rgba);
double ggo_iou(int n, const uint8_t* pred, const uint8_t* gt);
/* integer tallies behind metrics.evaluate / boundary_f1 / evaluate_trimap (reference metrics.py:58-201); out[14] */
void ggo_eval_counts(int h, int w, const uint8_t* pred, const uint8_t* gt, const uint8_t* trimap, int width, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif
