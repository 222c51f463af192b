/*
 * ggc.h — C ABI of libggc_hip.so, the MI355X (gfx950) implementation of the
 * GCN-GrabCut per-image segmentation hot path
 *     SLIC -> superpixel graph -> residual GCN -> guided-filter trimap -> GrabCut
 *
 * The reference (HanielUlises/GCN-GrabCut) has no FFI of its own: its boundary
 * is the Python API of src/gcn_grabcut.  Every entry point below states which
 * reference symbol (file:line under the reference tree) it replaces; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and shipped
 * in gcn-grabcut_amd/gcn_grabcut/_native.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - Pointers marked [dev] are device pointers on the context's GPU; the
 *     caller owns them.  [host] pointers are ordinary host memory.
 *   - Every call is batched over B images of identical H x W and is
 *     asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream) unless the comment says it synchronises.
 *   - Return value: GGC_OK or a negative GGC_E_* code; ggc_last_error() gives
 *     the message.  No C++ exception crosses the boundary.
 *   - A context is bound to one device and must be used by one host thread at
 *     a time.  Scratch memory grows monotonically inside the context and is
 *     released by ggc_ctx_destroy().
 *
 * Environment switches (all optional; the library reads them ONCE per process,
 * through one function, ggc::knobs() in csrc/ggc_context.hip; none changes a
 * result — tests/test_maxflow_variants_gpu.py holds the max-flow ones to that):
 *   GGC_MF_TRACE=1                per-round max-flow diagnostics on stderr (blocking)
 *   GGC_MF_WARM=0                 cold max-flow start in every GrabCut iteration (default 1: keep the n-link flow)
 *   GGC_MF_ASYNC=0                host-driven work lists only (default 1: sparse phases as one asynchronous launch each)
 *   GGC_MF_ASYNC_PUSH_ACTIVE=n    push rounds with <= n active pixels run asynchronously (10000)
 *   GGC_MF_ASYNC_TILE=8|16|32     rows of the asynchronous push tile (8)
 *   GGC_MF_ASYNC_HOPS=n           longest chain of tile visits in an asynchronous push launch (24)
 *   GGC_MF_ASYNC_SWEEPS=n         sweeps per asynchronous push visit (12)
 *   GGC_MF_DENSE_LAUNCHES0=n / GGC_MF_DENSE_LAUNCHES=n   push launches of the first / a later dense round (8 / 12)
 *   GGC_MF_DENSE_SWEEPS=n         sweeps per dense push visit (8)
 *   GGC_MF_RELAX_DENSE=n          work-list launches of a global relabel before the asynchronous launch takes over (3)
 *   GGC_MF_PARTIAL_ROUNDS=n       first rounds of a solve whose relabel stops after those launches (3; 0 = every relabel exact)
 *   GGC_AGG_DIRECT=1              GCNConv gather straight from L2 instead of the graph-resident kernel
 *   GGC_SLIC_SEQ_CONNECTIVITY=1   literal one-thread-per-image replay of skimage's connectivity pass (A/B reference)
 * The Python binding adds GGC_HIP_LIBRARY=<path> (load another build of the library, tools/build_variant.sh).
 */
#ifndef GGC_H
#define GGC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGC_VERSION 300 /* 0.3.0: round 3 — ggc_slic_rgb added, ggc_profile_enable(ctx, 2), one max-flow driver */

enum {
    GGC_OK            =  0,
    GGC_E_INVALID_ARG = -1,
    GGC_E_SHAPE       = -2,
    GGC_E_OOM         = -3,
    GGC_E_DEVICE      = -4,
    GGC_E_UNSUPPORTED = -5,
    GGC_E_STATE       = -6
};

/* GrabCut label space == cv2.GC_* == reference grabcut.py:22-27 (Label). */
enum { GGC_BGD = 0, GGC_FGD = 1, GGC_PR_BGD = 2, GGC_PR_FGD = 3 };

/* Feature widths: reference graph_builder.py:73-77. */
#define GGC_N_IMAGE_FEATS 16
#define GGC_N_PRIOR_FEATS 3
#define GGC_N_NODE_FEATS  19
#define GGC_N_EDGE_FEATS  5

typedef struct ggc_ctx ggc_ctx;
typedef void*          ggc_stream; /* hipStream_t */

/* ------------------------------------------------------------------ context */

int         ggc_version(void);
int         ggc_ctx_create(int device_id, ggc_ctx** out);
int         ggc_ctx_destroy(ggc_ctx* ctx);
const char* ggc_last_error(const ggc_ctx* ctx); /* ctx may be NULL: last create error */

/* Per-kernel timing for the roofline report (bench.py): while enabled, the
 * dominant kernels are bracketed by HIP events on their launch stream.
 * ggc_profile_enable(ctx, 1) clears earlier records and SYNCHRONISES the device;
 * ggc_profile_query sums the recorded durations of one kernel by name
 * ("gcn_aggregate", "gcn_gemm", "slic_assign", "maxflow", ...) and waits for them.
 * An event pair around nothing does not read zero (two queue packets), so enable
 * calibrates that offset with empty pairs and query subtracts it per scope; the
 * name "#event_pair_overhead" returns the offset itself (launches = 1).
 * on = 1: every instrumented scope; on = 2: only "gcn_aggregate", the kernel the
 * roofline grades (an event pair costs its stream ~10 us of idle time per scope, and a
 * step has several hundred scopes); on = 0: off. */
int ggc_profile_enable(ggc_ctx* ctx, int on);
int ggc_profile_query(ggc_ctx* ctx, const char* kernel, int* launches, double* total_ms);

/* Diagnostic hook for the parity tests: copy the first `bytes` of a named
 * scratch buffer ("slic_raw_labels", "slic_centers", "slic_image_a", ...) to
 * host memory.  SYNCHRONISES the device. */
int ggc_debug_read_scratch(ggc_ctx* ctx, const char* name, void* host_dst, size_t bytes);

/* --------------------------------------------------------------- G0 colour prep
 * Replaces GraphBuilder.__init__ (graph_builder.py:142-154): BGR->Lab (f64
 * arithmetic, stored f32), BGR->HSV (f64 -> f32), BGR->GRAY (8-bit fixed
 * point, stored f32), Sobel 3x3 gradient magnitude.
 *   bgr  [dev] u8  [B,H,W,3]
 *   lab  [dev] f32 [B,H,W,3]     hsv [dev] f32 [B,H,W,3]
 *   gray [dev] f32 [B,H,W]       grad [dev] f32 [B,H,W]
 */
int ggc_preprocess(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                   const uint8_t* bgr, float* lab, float* hsv, float* gray, float* grad);

/* --------------------------------------------------------------------- G1 SLIC
 * Replaces GraphBuilder._compute_superpixels (graph_builder.py:177-188), i.e.
 * skimage.segmentation.slic(lab_f32, n_segments, compactness, sigma,
 * start_label=0, channel_axis=-1): global min-max rescale (if rescale_input),
 * second rgb2lab in f32, Gaussian pre-smoothing, 10 k-means sweeps,
 * connectivity enforcement.  Labels are contiguous 0..n_nodes[b]-1.
 *   image    [dev] f32 [B,H,W,3]   (the Lab image from ggc_preprocess)
 *   segments [dev] i32 [B,H,W]
 *   n_nodes  [dev] i32 [B]
 */
int ggc_slic(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
             const float* image, int n_segments, float compactness, float sigma,
             int rescale_input, int32_t* segments, int32_t* n_nodes);

/* The same for SuperpixelGraphConfig(use_lab=False) (graph_builder.py:177-179): skimage.segmentation.slic on
 * `rgb.astype(float)`, i.e. the float64 instance of every stage (min-max rescale, rgb2lab, Gaussian, k-means).
 *   bgr      [dev] u8 [B,H,W,3]    (the image itself: the reference converts BGR -> RGB -> float64)
 */
int ggc_slic_rgb(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                 const uint8_t* bgr, int n_segments, double compactness, double sigma,
                 int32_t* segments, int32_t* n_nodes);

/* Step 7 of ggc_slic alone — skimage's _enforce_label_connectivity_cython
 * (SURVEY Appendix A.1): raw_labels, segments [dev] i32 [B,H,W]; n_nodes [dev] i32 [B].
 * SYNCHRONISES the stream once per carve round (components of >= max_size pixels). */
int ggc_slic_enforce_connectivity(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                                  const int32_t* raw_labels, int min_size, int max_size,
                                  int32_t* segments, int32_t* n_nodes);

/* ------------------------------------------------------------ G2-G8 graph build
 * Replaces GraphBuilder.build's tail (graph_builder.py:160-175):
 * _region_statistics, _assemble_node_features, _compute_edges (+_pair_features,
 * _nonlocal_pairs) and compute_auto_prior (graph_builder.py:357-444).
 *
 * Two-phase because N and E are data dependent.  ggc_graph_count runs the
 * whole construction into context scratch and returns the per-image offsets;
 * it SYNCHRONISES the stream once to read them back.  ggc_graph_fill copies
 * the result into caller buffers sized from those offsets.
 *   node_ptr [host] i64 [B+1]   prefix sums of n_nodes
 *   edge_ptr [host] i64 [B+1]   prefix sums of directed edge counts
 */
int ggc_graph_count(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                    const int32_t* segments, const int32_t* n_nodes,
                    const float* lab, const float* hsv, const float* grad,
                    int connectivity, int n_nonlocal,
                    int64_t* node_ptr, int64_t* edge_ptr);

/*   x          [dev] f32 [N_total,19]   node_input(): 16 image feats || 3 prior
 *   centroids  [dev] f32 [N_total,2]    (cy, cx) normalised
 *   area_ratio [dev] f32 [N_total]
 *   edge_src, edge_dst [dev] i32 [E_total]  LOCAL node ids (per image), in the
 *              reference's order: [adjacency pairs sorted, non-local pairs
 *              sorted] then the mirrored copy (graph_builder.py:303-306)
 *   edge_attr  [dev] f32 [E_total,5]
 * Any output pointer may be NULL to skip it.  global_ids != 0 adds each image's
 * node offset to the edge endpoints (PyG Batch collation), which is the form
 * ggc_resgcn_forward consumes.
 */
int ggc_graph_fill(ggc_ctx* ctx, ggc_stream stream,
                   float* x, float* centroids, float* area_ratio,
                   int32_t* edge_src, int32_t* edge_dst, float* edge_attr, int global_ids);

/* compute_auto_prior(segments, lab, centre_sigma=0.45, contrast_sigma=0.40) (graph_builder.py:357-362): the two sigmas of
 * the prior used by the NEXT ggc_graph_count calls of this context (defaults = the reference's). */
int ggc_graph_prior_sigmas(ggc_ctx* ctx, double centre_sigma, double contrast_sigma);

/* ------------------------------------------------------------ M0-M7 ResGCNNet
 * Replaces ResGCNNet (model.py:421-557), eval mode.
 * ggc_resgcn_configure fixes the architecture; ggc_resgcn_load_weight takes
 * one state_dict entry by its reference key, e.g.
 * "gcn_layers.3.lin.weight", as a contiguous f32 HOST array.  Integer buffers
 * ("...num_batches_tracked") are accepted and ignored.
 * hidden: any width from 8 to 128 (model.py:449-455 takes any); the entries are given in their TRUE shapes.  A width that is
 * not a multiple of 32 runs zero-padded to the next one inside the library (LayerNorm statistics on the true width).
 */
int ggc_resgcn_configure(ggc_ctx* ctx, int hidden, int n_layers);
int ggc_resgcn_load_weight(ggc_ctx* ctx, const char* name, const float* data /*[host]*/,
                           int64_t numel);
/* 0 when every tensor the forward pass needs has been loaded. */
int ggc_resgcn_ready(ggc_ctx* ctx);

/* Batched forward over G graphs (PyG Batch semantics, model.py:508-536):
 *   x         [dev] f32 [N,19]       edge_attr [dev] f32 [E,5]
 *   edge_src, edge_dst [dev] i32 [E] GLOBAL node ids (already offset per graph)
 *   node_ptr  [dev] i32 [G+1]        graph g owns nodes node_ptr[g]..node_ptr[g+1]
 *   logits    [dev] f32 [N,3]        (may be NULL)
 *   probs     [dev] f32 [N,3]        softmax(logits) (model.py:543-546; may be NULL)
 */
int ggc_resgcn_forward(ggc_ctx* ctx, ggc_stream stream, int G, int N, int E,
                       const float* x, const int32_t* edge_src, const int32_t* edge_dst,
                       const float* edge_attr, const int32_t* node_ptr,
                       float* logits, float* probs);

/* GCNTrimapNet, the reference's baseline model (`--model gcn`, model.py:239-316), eval mode.
 * Same protocol as the ResGCNNet entries: configure, load every float tensor of the state_dict by its key
 * ("in_norm.norm.weight", "blocks.0.conv.lin.weight", "blocks.0.edge_inject.proj.2.bias", "head.6.weight", ...;
 * *.num_batches_tracked is ignored), then forward.  The reference's forward takes no batch vector (the model has no
 * per-graph readout), so a batch is simply the concatenated graphs.
 *   x [dev] f32 [N,19]  edge_src, edge_dst [dev] i32 [E]  edge_attr [dev] f32 [E,5]
 *   logits, probs [dev] f32 [N,3] (either may be NULL) */
int ggc_gcnnet_configure(ggc_ctx* ctx, int hidden_channels, int n_layers);
int ggc_gcnnet_load_weight(ggc_ctx* ctx, const char* name, const float* data /*[host]*/, int64_t numel);
int ggc_gcnnet_ready(ggc_ctx* ctx);
int ggc_gcnnet_forward(ggc_ctx* ctx, ggc_stream stream, int N, int E,
                       const float* x, const int32_t* edge_src, const int32_t* edge_dst,
                       const float* edge_attr, float* logits, float* probs);

/* GATTrimapNet, the reference's attention variant (`--model gat`, model.py:323-414; SURVEY 8(f) last rank), eval mode: GATv2
 * attention with edge features (8 heads), LayerNorm + GELU, per-block edge gates, skip, per-graph attention readout, head.
 * Same protocol: configure, load every float tensor of the state_dict by its key ("convs.0.att", "convs.0.lin_l.weight",
 * "convs.0.lin_edge.weight", "lns.0.weight", "edge_gates.0.proj.2.bias", "skip_proj.weight", "ctx.attn.weight",
 * "head.3.weight", ...), then forward.  hidden_channels in {32, 64, 128}, n_heads == 8.  Arguments as ggc_resgcn_forward
 * (node_ptr delimits the graphs of the batch for the readout). */
int ggc_gat_configure(ggc_ctx* ctx, int hidden_channels, int n_heads, int n_layers);
int ggc_gat_load_weight(ggc_ctx* ctx, const char* name, const float* data /*[host]*/, int64_t numel);
int ggc_gat_ready(ggc_ctx* ctx);
int ggc_gat_forward(ggc_ctx* ctx, ggc_stream stream, int G, int N, int E,
                    const float* x, const int32_t* edge_src, const int32_t* edge_dst,
                    const float* edge_attr, const int32_t* node_ptr,
                    float* logits, float* probs);

/* M3 alone — the GCNConv scatter-gather the north star grades (PyG GCNConv
 * inside model.py:523-528).  CSR over destinations, self loops implicit.
 *   out_i = sum_{e: dst(e)=i} dis[src]*dis[i]*xw[src] + dis[i]*dis[i]*xw[i] + bias
 *   if gate != NULL:  h_out_i = h_i + gelu(out_i * gate_i)   (fused epilogue)
 *   else:             h_out_i = out_i
 *   xw [dev] f32 [N,D]  row_ptr [dev] i32 [N+1]  col [dev] i32 [E]
 *   dis [dev] f32 [N] = (1+indeg)^-1/2   bias [dev] f32 [D]
 */
int ggc_gcn_aggregate(ggc_ctx* ctx, ggc_stream stream, int N, int D,
                      const float* xw, const int32_t* row_ptr, const int32_t* col,
                      const float* dis, const float* bias,
                      const float* gate, const float* h, float* h_out);

/* Helper used with ggc_gcn_aggregate: build the destination CSR the forward
 * pass uses (stable in edge order) and dis = (1+indeg)^-1/2.
 *   row_ptr [dev] i32 [N+1]   col [dev] i32 [E]   dis [dev] f32 [N]
 */
int ggc_build_csr(ggc_ctx* ctx, ggc_stream stream, int N, int E,
                  const int32_t* edge_src, const int32_t* edge_dst,
                  int32_t* row_ptr, int32_t* col, float* dis);

/* ------------------------------------------------------------ P0-P3 trimap
 * Replaces refine_trimap (pipeline.py:103-146) incl. guided_filter
 * (pipeline.py:71-100) and project_to_pixels (model.py:648-661) when
 * edge_aware != 0; replaces _probs_to_trimap (model.py:664-678) otherwise.
 *   probs    [dev] f32 [N_total,3]     node_ptr [dev] i32 [B+1]
 *   segments [dev] i32 [B,H,W]         bgr [dev] u8 [B,H,W,3]
 *   trimap   [dev] u8  [B,H,W]  values in {0,1,2,3}
 */
int ggc_refine_trimap(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                      const float* probs, const int32_t* node_ptr,
                      const int32_t* segments, const uint8_t* bgr,
                      float threshold_fg, float threshold_bg,
                      int radius, float eps, int edge_aware, uint8_t* trimap);

/* P1 alone — replaces guided_filter (pipeline.py:71-100) for callers that use it
 * directly.  guide, src, out [dev] f32 [B,H,W]. */
int ggc_guided_filter(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                      const float* guide, const float* src, int radius, float eps, float* out);

/* S0 — replaces _seed_from_prior (pipeline.py:149-186); in place on trimap.
 *   prior [dev] f32 [N_total,3] (columns 16..18 of x, contiguous copy) */
int ggc_seed_from_prior(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                        const float* prior, const int32_t* node_ptr,
                        const int32_t* segments, double seed_frac, uint8_t* trimap);

/* ------------------------------------------------------------ C0-C6 GrabCut
 * Replaces GrabCut.run_with_trimap / run_with_bbox / refine
 * (grabcut.py:81-163), i.e. cv2.grabCut: GMM init (seeded k-means++), n_iter x
 * {assign components, learn GMMs, build graph, max-flow, relabel}.
 *   mode: 0 = GC_INIT_WITH_MASK (mask holds the trimap, promotions and the
 *             degenerate guard of grabcut.py:127-140 applied here),
 *         1 = GC_INIT_WITH_RECT (rects [host] i32 [B,4] = x,y,w,h),
 *         2 = GC_EVAL (reuse models)
 *   image  [dev] u8  [B,H,W,3]  (already in the configured colour space)
 *   mask   [dev] u8  [B,H,W]    in/out GrabCut labels
 *   bgd_model, fgd_model [dev] f64 [B,65]  in/out (5 coefs | 15 means | 45 covs)
 *   binary [dev] u8  [B,H,W]    out: mask in {FGD, PR_FGD} (grabcut.py:165-168)
 */
int ggc_grabcut(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                const uint8_t* image, uint8_t* mask, const int32_t* rects,
                double* bgd_model, double* fgd_model, int n_iter, int mode,
                uint64_t seed, uint8_t* binary);

/* K0 — replaces clean_mask (pipeline.py:189-227); 8-connected components.
 *   mask_in/mask_out [dev] u8 [B,H,W] in {0,1} (may alias) */
int ggc_clean_mask(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                   const uint8_t* mask_in, float min_area_ratio, int keep_largest,
                   uint8_t* mask_out);

/* O0 — replaces GrabCut.overlay_mask / crop_foreground (grabcut.py:180-195).
 *   overlay [dev] u8 [B,H,W,3]   rgba [dev] u8 [B,H,W,4]   (either may be NULL) */
int ggc_compose_outputs(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                        const uint8_t* bgr, const uint8_t* binary,
                        float alpha, int tint_b, int tint_g, int tint_r,
                        uint8_t* overlay, uint8_t* rgba);

/* R0 — IoU = tp / (tp + fp + fn + 1e-8) per image (metrics.py:79-84).
 *   iou [dev] f64 [B] (may be NULL)   counts [dev] u64 [B,3] = tp, fp, fn (may be NULL) */
int ggc_mask_iou(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                 const uint8_t* pred, const uint8_t* gt, double* iou, uint64_t* counts);

/* C7 — 8-bit colour spaces for GrabCutConfig.color_space (reference grabcut.py:73-79: cv2.cvtColor(BGR2HSV / BGR2Lab)
 * on uint8 images).  mode 0 = HSV (H in [0,180), OpenCV's fixed-point scheme), mode 1 = Lab
 * (L*255/100, a+128, b+128 of the float64 CIELAB, rounded).  OpenCV is absent here: parity with it is unpinned.
 *   bgr, out [dev] u8 [n_pixels,3] */
int ggc_convert_color8(ggc_ctx* ctx, ggc_stream stream, int64_t n_pixels, const uint8_t* bgr, int mode, uint8_t* out);

/* R1 — integer tallies behind metrics.evaluate / boundary_f1 / evaluate_trimap (metrics.py:58-129, 152-201), per
 * image.  counts [dev] u64 [B,14]:
 *   0 tp 1 fp 2 fn (pred / gt != 0)
 *   3 |pred boundary| 4 |gt boundary| 5 |both|: boundary = m - erode(m, ones(2*width+1)^2) with cv2.erode's default
 *     border (pixels outside the image never erode); zeros when boundary_width <= 0
 *   6 fg_tp 7 fg_fp 8 fg_fn 9 bg_tp 10 bg_fp 11 bg_fn 12 probable pixels 13 pixels where (FG|PR_FG) == gt value;
 *     zeros when trimap == NULL (trimap values as ggc_grabcut's mask). */
int ggc_eval_counts(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                    const uint8_t* pred, const uint8_t* gt, const uint8_t* trimap, int boundary_width,
                    uint64_t* counts);

/* D0 — per-region ground-truth coverage for the graph-cache writer: the integer sums behind
 * derive_trimap_labels and prepare_sample's fg_ratio (dataset.py:194-206, 245-248):
 *   counts[n] = pixels of region n,  fg[n] = pixels of region n with gt_mask > 0.
 *   segments [dev] i32 [B,H,W] (local labels)   gt_mask [dev] u8 [B,H,W]   node_ptr [dev] i32 [B+1]
 *   counts, fg [dev] i32 [node_ptr[B]]   (labels outside an image's node range are ignored) */
int ggc_region_label_stats(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                           const int32_t* segments, const uint8_t* gt_mask, const int32_t* node_ptr,
                           int32_t* counts, int32_t* fg);

#ifdef __cplusplus
}
#endif
#endif /* GGC_H */
