// ggc_internal.h — shared internals of libggc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "../../include/ggc.h"
#include "../../include/ggc_fmath.h"

namespace ggc {

constexpr int WAVE = 64;

// Scratch slots: one growing device buffer per slot, owned by the context.
enum Slot : int {
    S_CSR_ROWPTR = 0, S_CSR_COL, S_CSR_EID, S_CSR_CURSOR, S_CSR_SCAN, S_DIS, S_BATCH, S_AGG_PACK,
    S_STATES, S_GATE, S_XW, S_AGG, S_SCORE, S_GVEC, S_HJK,
    S_PRE_A, S_PRE_B, S_PRE_C, S_PRE_D,
    S_SLIC_IMG, S_SLIC_TMP, S_SLIC_CENTERS, S_SLIC_DIST, S_SLIC_LABELS, S_SLIC_AUX,
    S_SLIC_AUX2, S_SLIC_AUX3, S_SLIC_AUX4, S_SLIC_MINMAX,
    S_G_STATS, S_G_FEAT, S_G_PAIRS, S_G_PAIRCNT, S_G_NL, S_G_EDGE_SRC, S_G_EDGE_DST,
    S_G_EDGE_ATTR, S_G_PTR, S_G_PRIOR, S_G_AUX, S_G_AUX2, S_G_AUX3, S_G_X,
    S_T_A, S_T_B, S_T_C,
    S_GC_A, S_GC_B, S_GC_C, S_GC_D, S_GC_E, S_GC_F, S_GC_G, S_GC_H, S_GC_I, S_GC_J,
    S_GC_K, S_GC_L, S_GC_M, S_GC_N, S_GC_O,
    S_CC_A, S_CC_B, S_CC_C,
    S_MISC_A, S_MISC_B,
    S_COUNT
};

struct Buf {
    void*  p = nullptr;
    size_t bytes = 0;
};

// ResGCNNet parameters (reference model.py:449-499): host copies by state_dict
// key, device copies (raw + packed layouts) built lazily before a forward.
struct ResgcnWeights {
    int D = 0, n_layers = 0, Q = 0, C = 0;
    int Dt = 0;                            // true hidden width (ResGCNNet): D is Dt rounded up to a multiple of 32
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, Buf> dev;
    bool dev_ok = false;
};

// Result of ggc_graph_count kept until ggc_graph_fill.
struct GraphState {
    int B = 0, H = 0, W = 0;
    int64_t n_total = 0, e_total = 0;
    std::vector<int64_t> node_ptr, edge_ptr;
    int64_t max_pairs = 0;                 // largest per-image undirected pair count
    int n_max = 0;                         // row stride of the per-image scratch tables
    const int32_t* n_nodes_dev = nullptr;  // caller's n_nodes[B] (must outlive ggc_graph_fill)
    bool valid = false;
};

} // namespace ggc

namespace ggc {
// Optional per-kernel timing with HIP events on the launch stream (bench.py's
// roofline leg): off by default, costs two event records per profiled launch.
struct ProfRec { const char* name; hipEvent_t a, b; };
} // namespace ggc

struct ggc_ctx {
    int device = 0;
    std::string err;
    int prof_on = 0;                       // 0 off | 1 every scope | 2 the graded kernel's scope only
    double prof_overhead_ms = 0.0;         // duration an EMPTY event pair reports (calibrated at ggc_profile_enable)
    std::vector<ggc::ProfRec> prof;        // recorded scopes since ggc_profile_enable
    std::vector<hipEvent_t> prof_pool;     // recycled events
    ggc::Buf slots[ggc::S_COUNT];
    ggc::ResgcnWeights model;
    ggc::ResgcnWeights model2;             // GCNTrimapNet (same container: host copies by key, device copies)
    ggc::ResgcnWeights model3;             // GATTrimapNet (Q = attention heads)
    ggc::GraphState graph;
    int n_cu = 256;
    float prior_two_ce2 = (float)(2 * 0.45 * 0.45), prior_two_cs2 = (float)(2 * 0.40 * 0.40);   // compute_auto_prior sigmas (reference defaults)
    int32_t* h_pinned = nullptr;           // page-locked staging for small device -> host reads (no pageable-copy stall)
    static constexpr int H_PINNED_INTS = 4096;
};

namespace ggc {

// Every environment switch of the library, read ONCE per process by knobs() (ggc_context.hip); include/ggc.h documents them.
struct Knobs {
    int mf_trace;               // GGC_MF_TRACE: per-round max-flow diagnostics on stderr (blocking; tools/mf_trace.py)
    int mf_warm;                // GGC_MF_WARM (1): keep the n-link flow across GrabCut iterations
    int mf_async;               // GGC_MF_ASYNC (1): sparse max-flow phases as one asynchronous launch each; 0 = host-driven work lists only
    int mf_async_push_active;   // GGC_MF_ASYNC_PUSH_ACTIVE (10000): push rounds with at most this many active pixels run asynchronously
    int mf_async_tile;          // GGC_MF_ASYNC_TILE (8): rows of the asynchronous push tile, 8 | 16 | 32
    int mf_async_hops;          // GGC_MF_ASYNC_HOPS (24): longest chain of tile visits in an asynchronous push launch
    int mf_async_sweeps;        // GGC_MF_ASYNC_SWEEPS (12): sweeps per asynchronous push visit
    int mf_dense_launches0, mf_dense_launches;   // GGC_MF_DENSE_LAUNCHES0 (8) / GGC_MF_DENSE_LAUNCHES (12): push launches of the first / a later dense round
    int mf_dense_sweeps;        // GGC_MF_DENSE_SWEEPS (8): sweeps per dense push visit
    int mf_relax_dense;         // GGC_MF_RELAX_DENSE (3): work-list launches of a relabel before the asynchronous one takes over
    int mf_partial_rounds;      // GGC_MF_PARTIAL_ROUNDS (3): first rounds of a solve whose relabel stops after the work-list launches
    int agg_direct;             // GGC_AGG_DIRECT (0): 1 = GCNConv gather straight from L2 (k_aggregate) instead of the graph-resident kernel
    int slic_seq_connectivity;  // GGC_SLIC_SEQ_CONNECTIVITY (0): 1 = literal one-thread-per-image raster replay of skimage's connectivity pass
};
const Knobs& knobs();

int set_err(ggc_ctx* ctx, int code, const char* fmt, ...);
// small synchronous device -> host read through the context's pinned staging buffer (stream sync)
int read_i32(ggc_ctx* ctx, hipStream_t st, const int32_t* dev, int n, std::vector<int32_t>& host);
// Returns nullptr (and sets error) on failure. Content is NOT preserved on growth.
void* scratch(ggc_ctx* ctx, int slot, size_t bytes);

#define GGC_HIP(ctx, expr)                                                          \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess)                                                       \
            return ggc::set_err((ctx), GGC_E_DEVICE, "%s failed: %s (%s:%d)", #expr, \
                                hipGetErrorString(_e), __FILE__, __LINE__);          \
    } while (0)

#define GGC_LAUNCH_CHECK(ctx)                                                        \
    do {                                                                            \
        hipError_t _e = hipGetLastError();                                          \
        if (_e != hipSuccess)                                                       \
            return ggc::set_err((ctx), GGC_E_DEVICE, "kernel launch failed: %s (%s:%d)", \
                                hipGetErrorString(_e), __FILE__, __LINE__);          \
    } while (0)

#define GGC_REQUIRE(ctx, cond, code, ...)                      \
    do {                                                       \
        if (!(cond)) return ggc::set_err((ctx), (code), __VA_ARGS__); \
    } while (0)

template <typename T>
inline T* scratch_t(ggc_ctx* ctx, int slot, size_t count) {
    return reinterpret_cast<T*>(scratch(ctx, slot, count * sizeof(T)));
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute applies to the CURRENT device: "already done" is remembered per device (one bit each), because one
// process may hold contexts on several GPUs (cuda:0 then cuda:1).  Devices beyond 63 simply set the attribute every time.
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    bool need(int dev) const { return dev < 0 || dev >= 64 || !((mask.load(std::memory_order_acquire) >> dev) & 1ull); }
    void done(int dev) { if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release); }
};

struct ProfScope {
    ggc_ctx* ctx; hipStream_t st; hipEvent_t b = nullptr;
    ProfScope(ggc_ctx* c, hipStream_t s, const char* name) : ctx(c), st(s) {
        if (!c->prof_on || (c->prof_on == 2 && std::strcmp(name, "gcn_aggregate") != 0)) return;
        hipEvent_t ev[2];
        for (int i = 0; i < 2; ++i) {
            if (!c->prof_pool.empty()) { ev[i] = c->prof_pool.back(); c->prof_pool.pop_back(); }
            else if (hipEventCreate(&ev[i]) != hipSuccess) return;
        }
        (void)hipEventRecord(ev[0], s);
        b = ev[1];
        c->prof.push_back({name, ev[0], ev[1]});
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, st); }
};

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// exp / sigmoid / GELU of the networks: the fixed IEEE sequences of include/ggc_fmath.h, shared with the CPU oracle, so that
// the probabilities — and with them the trimap and the mask — come out bit-identical on both sides (libm, ocml and the
// hardware's v_exp_f32 / v_rcp_f32 each differ in the last ulp).  GELU(x) = x Phi(x) through an erfc fit, |error| 4e-7.
__device__ __forceinline__ float gelu_f(float x) { return ggc_geluf(x); }
__device__ __forceinline__ float sigmoid_f(float x) { return ggc_sigmoidf(x); }
// Four values at once: the polynomial parts on the packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32 are per-component IEEE
// operations), the exponent moves per component.  Same arithmetic as gelu_f, value for value.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f exp4_nonpos_f(v4f x) {      // ggc_expf for x <= 0 (the upper clamp cannot bind)
    x = __builtin_elementwise_max(x, (v4f)-87.0f);
    const v4f z = x * 1.44269504088896341f + 12582912.0f;
    const v4f n = z - 12582912.0f;
    v4f r = __builtin_elementwise_fma(n, (v4f)-0.693145751953125f, x);
    r = __builtin_elementwise_fma(n, (v4f)-1.42860682030941723212e-6f, r);
    v4f p = 1.3888889225e-3f;
    p = __builtin_elementwise_fma(p, r, (v4f)8.3333337680e-3f);
    p = __builtin_elementwise_fma(p, r, (v4f)4.1666667908e-2f);
    p = __builtin_elementwise_fma(p, r, (v4f)1.6666667163e-1f);
    p = __builtin_elementwise_fma(p, r, (v4f)0.5f);
    p = __builtin_elementwise_fma(p, r, (v4f)1.0f);
    p = __builtin_elementwise_fma(p, r, (v4f)1.0f);
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4u e = (__builtin_bit_cast(v4u, z) << 23) + 0x3F800000u;
    return p * __builtin_bit_cast(v4f, e);
}
__device__ __forceinline__ v4f gelu4_f(v4f x) {
    const v4f d = __builtin_elementwise_fma(__builtin_elementwise_abs(x), (v4f)0.27599036693573f, (v4f)1.0f);
    v4f u = __builtin_bit_cast(v4f, (v4i)0x7EF311C7 - __builtin_bit_cast(v4i, d));     // ggc_rcp_nr, four at a time
    u = u * __builtin_elementwise_fma(-d, u, (v4f)2.0f);
    u = u * __builtin_elementwise_fma(-d, u, (v4f)2.0f);
    u = u * __builtin_elementwise_fma(-d, u, (v4f)2.0f);
    v4f q = -0.11346635967493057f;
    q = __builtin_elementwise_fma(q, u, (v4f)0.44092419743537903f);
    q = __builtin_elementwise_fma(q, u, (v4f)-0.3140281140804291f);
    q = __builtin_elementwise_fma(q, u, (v4f)0.3222678005695343f);
    q = __builtin_elementwise_fma(q, u, (v4f)0.04667610302567482f);
    q = __builtin_elementwise_fma(q, u, (v4f)0.1176263764500618f);
    const v4f hq = (q * u) * exp4_nonpos_f((x * x) * -0.5f), om = 1.0f - hq;
    v4f phi;
    phi.x = x.x >= 0.0f ? om.x : hq.x; phi.y = x.y >= 0.0f ? om.y : hq.y;
    phi.z = x.z >= 0.0f ? om.z : hq.z; phi.w = x.w >= 0.0f ? om.w : hq.w;
    return x * phi;
}

// XCD-aware block remap: blocks b and b+8 share an XCD (observed round-robin
// placement; speed only).  Maps the hardware block id to a logical id so that
// each XCD walks one contiguous chunk of logical blocks.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;       // chunks: first r XCDs get q+1 blocks
    const int xcd = b & 7, slot = b >> 3;
    const int base = xcd * q + (xcd < r ? xcd : r);
    return base + slot;
}

} // namespace ggc
