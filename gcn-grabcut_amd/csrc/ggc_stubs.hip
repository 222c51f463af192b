// ggc_stubs.hip — entry points declared in include/ggc.h whose kernels are not
// written yet.  They fail loudly (GGC_E_UNSUPPORTED); there is no CPU fallback.
#include "ggc_internal.h"
#define STUB(ctx, name) return ggc::set_err((ctx), GGC_E_UNSUPPORTED, name " is not implemented yet")
extern "C" {
int ggc_grabcut(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, uint8_t*, const int32_t*, double*, double*, int, int, uint64_t, uint8_t*) { STUB(ctx, "ggc_grabcut"); }
int ggc_clean_mask(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, float, int, uint8_t*) { STUB(ctx, "ggc_clean_mask"); }
int ggc_compose_outputs(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, const uint8_t*, float, int, int, int, uint8_t*, uint8_t*) { STUB(ctx, "ggc_compose_outputs"); }
int ggc_mask_iou(ggc_ctx* ctx, ggc_stream, int, int, int, const uint8_t*, const uint8_t*, double*) { STUB(ctx, "ggc_mask_iou"); }
}

// Test/diagnostic hook: copy the head of a named scratch buffer to the host
// (synchronises).  Lets the parity tests localise a mismatch to a stage.
extern "C" int ggc_debug_read_scratch(ggc_ctx* ctx, const char* name, void* host_dst, size_t bytes) {
    if (!ctx || !name || !host_dst) return GGC_E_INVALID_ARG;
    static const struct { const char* n; int slot; } tab[] = {
        {"slic_image", -1}, {"slic_raw_labels", ggc::S_SLIC_LABELS}, {"slic_centers", ggc::S_SLIC_CENTERS},
        {"slic_image_a", ggc::S_SLIC_IMG}, {"slic_image_b", ggc::S_SLIC_TMP}, {"slic_stale", ggc::S_SLIC_AUX2},
    };
    for (auto& t : tab)
        if (t.slot >= 0 && std::strcmp(t.n, name) == 0) {
            const ggc::Buf& b = ctx->slots[t.slot];
            GGC_REQUIRE(ctx, b.p && b.bytes >= bytes, GGC_E_STATE, "scratch '%s' holds %zu bytes, asked for %zu", name,
                        b.bytes, bytes);
            GGC_HIP(ctx, hipSetDevice(ctx->device));
            GGC_HIP(ctx, hipDeviceSynchronize());
            GGC_HIP(ctx, hipMemcpy(host_dst, b.p, bytes, hipMemcpyDeviceToHost));
            return GGC_OK;
        }
    return ggc::set_err(ctx, GGC_E_INVALID_ARG, "unknown scratch name '%s'", name);
}
