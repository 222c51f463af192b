// ggc_post.hip — K0 mask clean-up, O0 output composition, R0 IoU.
//
// clean_mask (reference pipeline.py:189-227) needs 8-connected components with
// areas: a lock-free union-find over the foreground pixels (roots = smallest
// pixel index of a component, so numbering follows raster order like the CPU
// path), areas by integer atomics at the roots, then one pass applying the
// keep rule.  overlay_mask / crop_foreground: grabcut.py:180-195.
// IoU: metrics.py:79-84.
#include "ggc_internal.h"

namespace ggc {

struct PDims { int B, H, W, P; };

__device__ __forceinline__ int uf_find(const int32_t* parent, int i) {
    int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) { i = p; p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return i;
}
__device__ __forceinline__ void uf_union(int32_t* parent, int a, int b) {
    for (;;) {
        a = uf_find(parent, a); b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }      // link the larger root under the smaller
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

// Horizontal runs need no atomics: a pixel's parent starts as the first pixel of its run inside the wave's 64
// consecutive pixels (ballot of the run breaks); k_cc_merge only joins runs.
__global__ void __launch_bounds__(256) k_cc_init(PDims d, const uint8_t* __restrict__ mask, int32_t* __restrict__ parent,
                                                 int32_t* __restrict__ total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int on = 0, b = -1, p = 0;
    bool same_left = false;
    if (i < (size_t)d.B * d.P) {
        on = mask[i] != 0;
        b = (int)(i / d.P);
        p = (int)(i % d.P);
        same_left = on && (p % d.W) > 0 && mask[i - 1] != 0;
    }
    const unsigned long long starts = __ballot(on && (!same_left || lane == 0));
    if (i < (size_t)d.B * d.P)
        parent[i] = on ? p - (lane - (63 - __clzll((long long)(starts & ((2ull << lane) - 1ull))))) : -1;
    // total[b] != 0  <=>  the image has foreground (pipeline.py:208).  A wave can straddle images.
    if (d.P >= 64) {
        const int b0 = __builtin_amdgcn_readfirstlane(b);
        const unsigned long long same = __ballot(on && b == b0), next = __ballot(on && b != b0);
        if ((threadIdx.x & 63) == 0 && b0 >= 0) {
            if (same) atomicOr(&total[b0], 1);
            if (next) atomicOr(&total[b0 + 1], 1);
        }
    } else if (on) {
        atomicOr(&total[b], 1);
    }
}

__global__ void __launch_bounds__(256) k_cc_merge(PDims d, int32_t* __restrict__ parent) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= d.W || y >= d.H) return;
    const size_t base = (size_t)blockIdx.z * d.P;
    int32_t* par = parent + base;
    const int p = y * d.W + x;
    if (par[p] < 0) return;
    // 8-connectivity with the already-scanned neighbours (left, up-left, up, up-right).  Foreground pixels of one row
    // that touch are one run already, so a pixel whose left neighbour is foreground only adds what that neighbour
    // could not see: the up-right pixel when the pixel above is background.
    const bool left = x > 0 && par[p - 1] >= 0;
    if (left && ((base + p) & 63) == 0) uf_union(par, p, p - 1);                       // run continues across k_cc_init's wave boundary
    if (y > 0) {
        const bool up = par[p - d.W] >= 0;
        const bool ul = x > 0 && par[p - d.W - 1] >= 0, ur = x + 1 < d.W && par[p - d.W + 1] >= 0;
        if (left) {
            if (ur && !up) uf_union(par, p, p - d.W + 1);
        } else if (up) {
            uf_union(par, p, p - d.W);
        } else {
            if (ul) uf_union(par, p, p - d.W - 1);
            if (ur) uf_union(par, p, p - d.W + 1);
        }
    }
}

// NB: a wave of a row-block may straddle two images only if P % 64 != 0; total[] is per image so guard by image.
__global__ void __launch_bounds__(256) k_cc_area(PDims d, int32_t* __restrict__ parent, int32_t* __restrict__ area) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < (size_t)d.B * d.P && parent[i] >= 0;
    bool same_left = false;
    size_t base = 0;
    int r = 0;
    if (valid) {
        base = (i / d.P) * d.P;
        const int p = (int)(i - base);
        same_left = (p % d.W) > 0 && parent[i - 1] >= 0;
        r = uf_find(parent + base, p);
        parent[i] = r;                                      // compress (roots keep pointing at themselves)
    }
    // one atomic per run (the runs of k_cc_init), not per pixel
    const unsigned long long vmask = __ballot(valid);
    const unsigned long long starts = __ballot(valid && (!same_left || lane == 0));
    if (valid && ((starts >> lane) & 1ull)) {
        const unsigned long long stop = (starts | ~vmask) >> lane >> 1;
        const int len = stop ? __ffsll((long long)stop) : 64 - lane;
        atomicAdd(&area[base + r], len);
    }
}

// best[b] = max over components of (area << 32 | ~root): largest area, ties to the first component in raster order
__global__ void __launch_bounds__(256) k_cc_best(PDims d, double min_area, const int32_t* __restrict__ parent,
                                                 const int32_t* __restrict__ area, unsigned long long* __restrict__ best,
                                                 int32_t* __restrict__ any) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    const int b = (int)(i / d.P), p = (int)(i % d.P);
    if (parent[i] != p) return;                              // roots only
    const int a = area[i];
    atomicMax(&best[b], ((unsigned long long)(unsigned)a << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p));
    if ((double)a >= min_area) atomicOr(&any[b], 1);
}

__global__ void __launch_bounds__(256) k_cc_apply(PDims d, double min_area, int keep_largest, int passthrough,
                                                  const uint8_t* __restrict__ mask, const int32_t* __restrict__ parent,
                                                  const int32_t* __restrict__ area,
                                                  const unsigned long long* __restrict__ best,
                                                  const int32_t* __restrict__ any, const int32_t* __restrict__ total,
                                                  uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    const int b = (int)(i / d.P);
    if (passthrough || total[b] == 0) { out[i] = mask[i]; return; }   // pipeline.py:208-209
    const int r = parent[i];
    uint8_t keep = 0;
    if (r >= 0) {
        const size_t base = (size_t)b * d.P;
        const int best_root = (int)(0xFFFFFFFFu - (unsigned)(best[b] & 0xFFFFFFFFull));
        keep = (keep_largest || !any[b]) ? (r == best_root) : ((double)area[base + r] >= min_area);
    }
    out[i] = keep;
}

__global__ void __launch_bounds__(256) k_compose(size_t n, const uint8_t* __restrict__ bgr, const uint8_t* __restrict__ binary,
                                                 float alpha, float tb, float tg, float tr, uint8_t* __restrict__ overlay,
                                                 uint8_t* __restrict__ rgba) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float m = binary[i] ? 1.0f : 0.0f;
    const float tint[3] = {tb, tg, tr};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint8_t v8 = bgr[3 * i + c];
        if (overlay) {
            float v = (float)v8 * (1.0f - alpha * m) + (tint[c] * alpha) * m;
            v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
            overlay[3 * i + c] = (uint8_t)v;
        }
        if (rgba) rgba[4 * i + c] = v8;
    }
    if (rgba) rgba[4 * i + 3] = binary[i] ? 255 : 0;
}

__global__ void __launch_bounds__(256) k_iou_count(PDims d, const uint8_t* __restrict__ pred, const uint8_t* __restrict__ gt,
                                                   unsigned long long* __restrict__ cnt /*[B,3] tp fp fn*/) {
    const int b = blockIdx.y;
    unsigned int tp = 0, fp = 0, fn = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < d.P; p += gridDim.x * blockDim.x) {
        const int a = pred[(size_t)b * d.P + p] != 0, g = gt[(size_t)b * d.P + p] != 0;
        tp += a & g; fp += a & (1 - g); fn += (1 - a) & g;
    }
    for (int o = 32; o > 0; o >>= 1) { tp += __shfl_xor(tp, o, 64); fp += __shfl_xor(fp, o, 64); fn += __shfl_xor(fn, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&cnt[3 * b], (unsigned long long)tp); atomicAdd(&cnt[3 * b + 1], (unsigned long long)fp);
        atomicAdd(&cnt[3 * b + 2], (unsigned long long)fn);
    }
}
__global__ void k_iou_final(int B, const unsigned long long* __restrict__ cnt, double* __restrict__ iou,
                            uint64_t* __restrict__ counts) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double tp = (double)cnt[3 * b], fp = (double)cnt[3 * b + 1], fn = (double)cnt[3 * b + 2];
    if (iou) iou[b] = tp / ((tp + fp + fn) + 1e-8);
    if (counts) { counts[3 * b] = cnt[3 * b]; counts[3 * b + 1] = cnt[3 * b + 1]; counts[3 * b + 2] = cnt[3 * b + 2]; }
}

} // namespace ggc

using namespace ggc;

extern "C" int ggc_clean_mask(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const uint8_t* mask_in,
                              float min_area_ratio, int keep_largest, uint8_t* mask_out) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, mask_in && mask_out, GGC_E_INVALID_ARG, "null pointer");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PDims d{B, H, W, H * W};
    const size_t BP = (size_t)B * d.P;
    int32_t* parent = scratch_t<int32_t>(ctx, S_CC_A, BP);
    int32_t* area = scratch_t<int32_t>(ctx, S_CC_B, BP);
    unsigned long long* best = scratch_t<unsigned long long>(ctx, S_CC_C, (size_t)B * 2);
    if (!parent || !area || !best) return GGC_E_OOM;
    int32_t* any = reinterpret_cast<int32_t*>(best + B);
    int32_t* total = any + B;
    const int passthrough = (min_area_ratio <= 0.0f && !keep_largest) ? 1 : 0;
    const double min_area = (double)min_area_ratio * (double)d.P;
    GGC_HIP(ctx, hipMemsetAsync(best, 0, sizeof(unsigned long long) * B * 2, st));
    GGC_HIP(ctx, hipMemsetAsync(area, 0, sizeof(int32_t) * BP, st));
    const dim3 g1(cdiv(BP, 256));
    hipLaunchKernelGGL(k_cc_init, g1, dim3(256), 0, st, d, mask_in, parent, total);
    if (!passthrough) {
        hipLaunchKernelGGL(k_cc_merge, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(256), 0, st, d, parent);
        hipLaunchKernelGGL(k_cc_area, g1, dim3(256), 0, st, d, parent, area);
        hipLaunchKernelGGL(k_cc_best, g1, dim3(256), 0, st, d, min_area, parent, area, best, any);
    }
    hipLaunchKernelGGL(k_cc_apply, g1, dim3(256), 0, st, d, min_area, keep_largest, passthrough, mask_in, parent, area, best,
                       any, total, mask_out);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

extern "C" int ggc_compose_outputs(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const uint8_t* bgr,
                                   const uint8_t* binary, float alpha, int tint_b, int tint_g, int tint_r,
                                   uint8_t* overlay, uint8_t* rgba) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, bgr && binary, GGC_E_INVALID_ARG, "null pointer");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(k_compose, dim3(cdiv(n, 256)), dim3(256), 0, st, n, bgr, binary, alpha, (float)tint_b, (float)tint_g,
                       (float)tint_r, overlay, rgba);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

extern "C" int ggc_mask_iou(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const uint8_t* pred,
                            const uint8_t* gt, double* iou, uint64_t* counts) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, pred && gt && (iou || counts), GGC_E_INVALID_ARG, "null pointer");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PDims d{B, H, W, H * W};
    unsigned long long* cnt = scratch_t<unsigned long long>(ctx, S_CC_C, (size_t)B * 3);
    if (!cnt) return GGC_E_OOM;
    GGC_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * B * 3, st));
    hipLaunchKernelGGL(k_iou_count, dim3(std::min(cdiv(d.P, 1024), 128), B), dim3(256), 0, st, d, pred, gt, cnt);
    hipLaunchKernelGGL(k_iou_final, dim3(cdiv(B, 64)), dim3(64), 0, st, B, cnt, iou, counts);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

// ------------------------------------------------------------------ D0: region label statistics
namespace ggc {
// One wave covers 64 consecutive pixels of an image; the pixels of one region inside it form runs, and a run adds its
// length (and its foreground count) with one atomic each — integer sums, exact in any order.
__global__ void __launch_bounds__(256) k_region_label_stats(PDims d, const int32_t* __restrict__ seg, const uint8_t* __restrict__ gt,
                                                            const int32_t* __restrict__ node_ptr, int32_t* __restrict__ counts,
                                                            int32_t* __restrict__ fg) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int n0 = node_ptr[b], n = node_ptr[b + 1] - n0;
    for (int p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; p0 < d.P; p0 += gridDim.x * 256) {
        const int p = p0 + lane;
        int s = -1, on = 0;
        if (p < d.P) {
            s = seg[(size_t)b * d.P + p];
            on = gt[(size_t)b * d.P + p] > 0;
            if (s < 0 || s >= n) s = -1;
        }
        const int s_left = __shfl_up(s, 1, 64);
        const unsigned long long starts = __ballot(lane == 0 || s_left != s);
        const unsigned long long fgm = __ballot(on != 0);
        if (s >= 0 && ((starts >> lane) & 1ull)) {
            const unsigned long long stop = starts >> lane >> 1;
            const int len = stop ? __ffsll((long long)stop) : 64 - lane;
            const unsigned long long run = (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
            atomicAdd(&counts[n0 + s], len);
            const int nf = __popcll(fgm & run);
            if (nf) atomicAdd(&fg[n0 + s], nf);
        }
    }
}
} // namespace ggc

extern "C" int ggc_region_label_stats(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const int32_t* segments,
                                      const uint8_t* gt_mask, const int32_t* node_ptr, int32_t* counts, int32_t* fg) {
    using namespace ggc;
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, segments && gt_mask && node_ptr && counts && fg, GGC_E_INVALID_ARG, "null pointer");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PDims d{B, H, W, H * W};
    std::vector<int32_t> total;
    int rc = read_i32(ctx, st, node_ptr + B, 1, total);
    if (rc) return rc;
    GGC_REQUIRE(ctx, total[0] >= 0, GGC_E_INVALID_ARG, "node_ptr[B] = %d", total[0]);
    if (total[0] > 0) {
        GGC_HIP(ctx, hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)total[0], st));
        GGC_HIP(ctx, hipMemsetAsync(fg, 0, sizeof(int32_t) * (size_t)total[0], st));
    }
    hipLaunchKernelGGL(k_region_label_stats, dim3(std::min(cdiv(d.P, 256), 256), B), dim3(256), 0, st, d, segments, gt_mask, node_ptr,
                       counts, fg);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

// ------------------------------------------------------------------ R1: evaluation tallies
namespace ggc {
constexpr int EVAL_N = 14;

__device__ __forceinline__ bool eval_boundary(const PDims& d, const uint8_t* __restrict__ m, int y, int x, int width) {
    if (!m[(size_t)y * d.W + x]) return false;
    const int y0 = max(y - width, 0), y1 = min(y + width, d.H - 1), x0 = max(x - width, 0), x1 = min(x + width, d.W - 1);
    for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx)
            if (!m[(size_t)yy * d.W + xx]) return true;      // a background pixel inside the window erodes this one
    return false;
}

__global__ void __launch_bounds__(256) k_eval_counts(PDims d, const uint8_t* __restrict__ pred, const uint8_t* __restrict__ gt,
                                                     const uint8_t* __restrict__ trimap, int width,
                                                     unsigned long long* __restrict__ counts) {
    __shared__ unsigned int s_cnt[EVAL_N];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid < EVAL_N) s_cnt[tid] = 0;
    __syncthreads();
    const uint8_t* pb = pred + (size_t)b * d.P;
    const uint8_t* gb = gt + (size_t)b * d.P;
    unsigned int c[EVAL_N];
#pragma unroll
    for (int i = 0; i < EVAL_N; ++i) c[i] = 0;
    for (int p = blockIdx.x * 256 + tid; p < d.P; p += gridDim.x * 256) {
        const int y = p / d.W, x = p - y * d.W;
        const bool pp = pb[p] != 0, g = gb[p] != 0;
        c[0] += pp && g; c[1] += pp && !g; c[2] += !pp && g;
        if (width > 0) {
            const bool bp = eval_boundary(d, pb, y, x, width), bg = eval_boundary(d, gb, y, x, width);
            c[3] += bp; c[4] += bg; c[5] += bp && bg;
        }
        if (trimap) {
            const int t = trimap[(size_t)b * d.P + p];
            const bool pf = t == GGC_FGD, pbg = t == GGC_BGD;
            c[6] += pf && g; c[7] += pf && !g; c[8] += !pf && g;
            c[9] += pbg && !g; c[10] += pbg && g; c[11] += !pbg && !g;
            c[12] += t == GGC_PR_BGD || t == GGC_PR_FGD;
            c[13] += ((t == GGC_FGD || t == GGC_PR_FGD) ? 1 : 0) == (int)gb[p];
        }
    }
#pragma unroll
    for (int i = 0; i < EVAL_N; ++i) {
        unsigned int v = c[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 63) == 0 && v) atomicAdd(&s_cnt[i], v);
    }
    __syncthreads();
    if (tid < EVAL_N && s_cnt[tid]) atomicAdd(&counts[(size_t)b * EVAL_N + tid], (unsigned long long)s_cnt[tid]);
}
} // namespace ggc

extern "C" int ggc_eval_counts(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const uint8_t* pred, const uint8_t* gt,
                               const uint8_t* trimap, int boundary_width, uint64_t* counts) {
    using namespace ggc;
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, pred && gt && counts, GGC_E_INVALID_ARG, "null pointer");
    GGC_REQUIRE(ctx, boundary_width <= 64, GGC_E_INVALID_ARG, "boundary width %d out of range", boundary_width);
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PDims d{B, H, W, H * W};
    GGC_HIP(ctx, hipMemsetAsync(counts, 0, sizeof(uint64_t) * (size_t)B * EVAL_N, st));
    hipLaunchKernelGGL(k_eval_counts, dim3(std::min(cdiv(d.P, 256), 128), B), dim3(256), 0, st, d, pred, gt, trimap, boundary_width,
                       reinterpret_cast<unsigned long long*>(counts));
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}
