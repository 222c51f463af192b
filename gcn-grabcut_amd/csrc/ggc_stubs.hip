// ggc_stubs.hip — entry points declared in include/ggc.h whose kernels are not
// written yet.  They fail loudly (GGC_E_UNSUPPORTED); there is no CPU fallback.
#include "ggc_internal.h"
#define STUB(ctx, name) return ggc::set_err((ctx), GGC_E_UNSUPPORTED, name " is not implemented yet")
extern "C" {
int ggc_preprocess(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, float*, float*, float*, float*) { STUB(ctx, "ggc_preprocess"); }
int ggc_slic(ggc_ctx* ctx, ggc_stream, int, int, int, const float*, int, float, float, int, int32_t*, int32_t*) { STUB(ctx, "ggc_slic"); }
int ggc_graph_count(ggc_ctx* ctx, ggc_stream, int, int, int, const int32_t*, const int32_t*, const float*, const float*, const float*, int, int, int64_t*, int64_t*) { STUB(ctx, "ggc_graph_count"); }
int ggc_graph_fill(ggc_ctx* ctx, ggc_stream, float*, float*, float*, int32_t*, int32_t*, float*) { STUB(ctx, "ggc_graph_fill"); }
int ggc_refine_trimap(ggc_ctx* ctx, ggc_stream, int, int, int, const float*, const int32_t*, const int32_t*, const uint8_t*, float, float, int, float, int, uint8_t*) { STUB(ctx, "ggc_refine_trimap"); }
int ggc_seed_from_prior(ggc_ctx* ctx, ggc_stream, int, int, int, const float*, const int32_t*, const int32_t*, float, uint8_t*) { STUB(ctx, "ggc_seed_from_prior"); }
int ggc_grabcut(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, uint8_t*, const int32_t*, double*, double*, int, int, uint64_t, uint8_t*) { STUB(ctx, "ggc_grabcut"); }
int ggc_clean_mask(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, float, int, uint8_t*) { STUB(ctx, "ggc_clean_mask"); }
int ggc_compose_outputs(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, const uint8_t*, float, int, int, int, uint8_t*, uint8_t*) { STUB(ctx, "ggc_compose_outputs"); }
int ggc_mask_iou(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, const uint8_t*, double*) { STUB(ctx, "ggc_mask_iou"); }
}
