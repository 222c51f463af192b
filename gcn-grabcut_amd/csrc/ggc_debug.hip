// ggc_debug.hip — diagnostic hook used by the parity tests (no compute).
#include "ggc_internal.h"

// Copy the head of a named scratch buffer to the host (synchronises).  Lets the
// parity tests localise a mismatch to a stage.
extern "C" int ggc_debug_read_scratch(ggc_ctx* ctx, const char* name, void* host_dst, size_t bytes) {
    if (!ctx || !name || !host_dst) return GGC_E_INVALID_ARG;
    static const struct { const char* n; int slot; } tab[] = {
        {"slic_raw_labels", ggc::S_SLIC_LABELS}, {"slic_centers", ggc::S_SLIC_CENTERS},
        {"slic_image_a", ggc::S_SLIC_IMG}, {"slic_image_b", ggc::S_SLIC_TMP}, {"slic_stale", ggc::S_SLIC_AUX2},
        {"grabcut_comp", ggc::S_GC_D}, {"grabcut_nweights", ggc::S_GC_E}, {"grabcut_gmm", ggc::S_GC_B},
        {"grabcut_excess_sink_dist", ggc::S_GC_G},
        {"net_states", ggc::S_STATES}, {"net_xw", ggc::S_XW}, {"net_agg", ggc::S_AGG}, {"net_hjk", ggc::S_HJK},
        {"net_score", ggc::S_SCORE}, {"net_gvec", ggc::S_GVEC}, {"net_gate", ggc::S_GATE},
    };
    for (auto& t : tab)
        if (std::strcmp(t.n, name) == 0) {
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
