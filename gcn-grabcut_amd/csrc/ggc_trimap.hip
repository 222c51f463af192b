// ggc_trimap.hip — P0-P3 and S0: region probabilities -> pixel trimap.
//
// Replaces refine_trimap (reference pipeline.py:103-146) with guided_filter
// (pipeline.py:71-100) and project_to_pixels (model.py:648-661); the
// non-edge-aware variant _probs_to_trimap (model.py:664-678); and
// _seed_from_prior (pipeline.py:149-186).
//
// The trimap is an integer output that must equal the CPU path bit for bit, so
// the box filter is defined (here and in the oracle) as float64 sums in a fixed
// order — the 2r+1 taps of a row left to right, then the 2r+1 row sums top to
// bottom, times 1/(2r+1)^2, cast to float32 — with BORDER_REFLECT_101, and all
// other arithmetic is float32 without FMA contraction.  Direct (non-sliding)
// sums make every output independent of its neighbours' rounding history.
//
// HBM layout: planes [plane][B][H*W]; the six first-stage planes (g, s_bg, s_fg,
// g*g, g*s_bg, g*s_fg) are blurred together, then the four (a, b) planes.  The
// guide statistics are shared between the BG and FG filters (10 blurs, not 14).
#include "ggc_internal.h"
#include <cmath>

namespace ggc {

__device__ __forceinline__ int refl101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; }
    return i;
}

struct TDims { int B, H, W; };

// guide = gray/255, projected probabilities and their products (6 planes)
__global__ void __launch_bounds__(256) k_t_prep(TDims d, const float* __restrict__ probs,
                                                const int32_t* __restrict__ node_ptr,
                                                const int32_t* __restrict__ seg, const uint8_t* __restrict__ bgr,
                                                float* __restrict__ planes) {
    const size_t P = (size_t)d.H * d.W, BP = P * d.B;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BP) return;
    const int b = (int)(i / P);
    const uint8_t* px = bgr + i * 3;
    const int g8 = (px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + (1 << 14)) >> 15;
    const float g = (float)g8 / (float)255.0;
    const int n0 = node_ptr[b], n = node_ptr[b + 1] - n0;
    const int s = seg[i];
    float pb = 0.0f, pf = 0.0f;                 // project_to_pixels zero-pads missing regions
    if (s >= 0 && s < n) { pb = probs[(size_t)(n0 + s) * 3 + 0]; pf = probs[(size_t)(n0 + s) * 3 + 2]; }
    planes[i] = g;
    planes[BP + i] = pb;
    planes[2 * BP + i] = pf;
    planes[3 * BP + i] = g * g;
    planes[4 * BP + i] = g * pb;
    planes[5 * BP + i] = g * pf;
}

// horizontal pass: f64 row sums
__global__ void __launch_bounds__(256) k_blur_h(TDims d, int n_planes, int radius, const float* __restrict__ in,
                                                double* __restrict__ hs) {
    const size_t P = (size_t)d.H * d.W;
    const size_t total = P * d.B * n_planes;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % d.W);
    const float* row = in + (i - x);
    double s = 0.0;
    for (int dx = -radius; dx <= radius; ++dx) s += (double)row[refl101(x + dx, d.W)];
    hs[i] = s;
}

// vertical pass: f64 column sums of the row sums, scale, cast
__global__ void __launch_bounds__(256) k_blur_v(TDims d, int n_planes, int radius, const double* __restrict__ hs,
                                                float* __restrict__ out) {
    const size_t P = (size_t)d.H * d.W;
    const size_t total = P * d.B * n_planes;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t img = i / P;
    const int p = (int)(i % P);
    const int y = p / d.W, x = p - y * d.W;
    const double* base = hs + img * P;
    const int k = 2 * radius + 1;
    const double scale = 1.0 / ((double)k * (double)k);
    double s = 0.0;
    for (int dy = -radius; dy <= radius; ++dy) s += base[(size_t)refl101(y + dy, d.H) * d.W + x];
    out[i] = (float)(s * scale);
}

// Both passes in one kernel for radius 1..8 (the pipeline uses 8): a block owns a 64 x 32 output tile of one plane,
// stages the tile + halo in LDS (reflected coordinates applied at load), builds the f64 row sums of its 48 rows in
// LDS and then the column sums.  Same taps in the same order as k_blur_h + k_blur_v — every sum is a fresh 2r+1-term
// sum — so the result is bit-identical, but each plane is read and written once instead of crossing HBM as f64 row
// sums in between (20 B/px -> 8 B/px, and no strided second pass).
constexpr int BX_W = 64, BX_H = 32, BX_R = 8;

template <int radius>
__global__ void __launch_bounds__(256) k_box_fused(TDims d, const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float s_in[BX_H + 2 * radius][BX_W + 2 * radius + 1];
    __shared__ double s_rs[BX_H + 2 * radius][BX_W];
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * BX_W, ty0 = blockIdx.y * BX_H;
    const size_t P = (size_t)d.H * d.W;
    const float* src = in + (size_t)blockIdx.z * P;
    constexpr int iw = BX_W + 2 * radius, ih = BX_H + 2 * radius;
    for (int i = tid; i < ih * iw; i += 256) {
        const int ly = i / iw, lx = i - ly * iw;
        // rows / columns past the image edge are never used by an in-image output, clamp them into range first
        const int gy = refl101(min(ty0 + ly - radius, d.H - 1 + radius), d.H);
        const int gx = refl101(min(tx0 + lx - radius, d.W - 1 + radius), d.W);
        s_in[ly][lx] = src[(size_t)gy * d.W + gx];
    }
    __syncthreads();
    // row sums: 4 adjacent outputs per work item share their 2r+4 taps
    for (int w = tid; w < ih * (BX_W / 4); w += 256) {
        const int ly = w / (BX_W / 4), x4 = (w - ly * (BX_W / 4)) * 4;
        double v[2 * radius + 4];
#pragma unroll
        for (int j = 0; j < 2 * radius + 4; ++j) v[j] = (double)s_in[ly][x4 + j];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j <= 2 * radius; ++j) sum += v[o + j];
            s_rs[ly][x4 + o] = sum;
        }
    }
    __syncthreads();
    const int k = 2 * radius + 1;
    const double scale = 1.0 / ((double)k * (double)k);
    float* dst = out + (size_t)blockIdx.z * P;
    // column sums: 8 adjacent rows of one column per work item
    for (int w = tid; w < (BX_H / 8) * BX_W; w += 256) {
        const int x = w % BX_W, y8 = (w / BX_W) * 8;
        double v[2 * radius + 8];
#pragma unroll
        for (int j = 0; j < 2 * radius + 8; ++j) v[j] = s_rs[y8 + j][x];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j <= 2 * radius; ++j) sum += v[o + j];
            const int gy = ty0 + y8 + o, gx = tx0 + x;
            if (gy < d.H && gx < d.W) dst[(size_t)gy * d.W + gx] = (float)(sum * scale);
        }
    }
}

// box filter of n_planes [B][H*W] planes: fused for radius <= 8, the two HBM passes otherwise
static void box_filter(hipStream_t st, const TDims& d, int n_planes, int radius, const float* in, double* hs, float* out) {
    const size_t BP = (size_t)d.B * d.H * d.W;
    const dim3 grid(cdiv(d.W, BX_W), cdiv(d.H, BX_H), d.B * n_planes);
    switch ((size_t)d.B * n_planes <= 65535 ? radius : -1) {     // grid.z limit: very large batches take the two-pass route
#define GGC_BOX(R) case R: hipLaunchKernelGGL(k_box_fused<R>, grid, dim3(256), 0, st, d, in, out); return;
        GGC_BOX(1) GGC_BOX(2) GGC_BOX(3) GGC_BOX(4) GGC_BOX(5) GGC_BOX(6) GGC_BOX(7) GGC_BOX(8)
#undef GGC_BOX
        default: break;
    }
    hipLaunchKernelGGL(k_blur_h, dim3(cdiv(BP * n_planes, 256)), dim3(256), 0, st, d, n_planes, radius, in, hs);
    hipLaunchKernelGGL(k_blur_v, dim3(cdiv(BP * n_planes, 256)), dim3(256), 0, st, d, n_planes, radius, hs, out);
}

// a = cov / (var + eps), b = mean_s - a * mean_g for BG and FG (4 planes)
__global__ void __launch_bounds__(256) k_t_ab(size_t BP, float eps, const float* __restrict__ m /*6 planes*/,
                                              float* __restrict__ ab /*4 planes*/) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BP) return;
    const float mg = m[i], msb = m[BP + i], msf = m[2 * BP + i];
    const float var = m[3 * BP + i] - mg * mg;
    const float cov_b = m[4 * BP + i] - mg * msb;
    const float cov_f = m[5 * BP + i] - mg * msf;
    const float a_b = cov_b / (var + eps), a_f = cov_f / (var + eps);
    ab[i] = a_b;
    ab[BP + i] = msb - a_b * mg;
    ab[2 * BP + i] = a_f;
    ab[3 * BP + i] = msf - a_f * mg;
}

__device__ __forceinline__ uint8_t decide(float b, float f, float thr_bg, float thr_fg) {
    uint8_t lab = f > b ? GGC_PR_FGD : GGC_PR_BGD;
    if (b >= thr_bg) lab = GGC_BGD;
    if (f >= thr_fg) lab = GGC_FGD;          // FG wins (pipeline.py:143-145)
    return lab;
}

__global__ void __launch_bounds__(256) k_t_final(size_t BP, float thr_fg, float thr_bg,
                                                 const float* __restrict__ guide, const float* __restrict__ mab,
                                                 uint8_t* __restrict__ trimap) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BP) return;
    const float g = guide[i];
    float b = mab[i] * g + mab[BP + i];
    float f = mab[2 * BP + i] * g + mab[3 * BP + i];
    b = b < 0.0f ? 0.0f : (b > 1.0f ? 1.0f : b);
    f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
    trimap[i] = decide(b, f, thr_bg, thr_fg);
}

// non-edge-aware path: threshold region probabilities, pad with PR_BGD, gather
__global__ void __launch_bounds__(256) k_t_plain(TDims d, float thr_fg, float thr_bg, const float* __restrict__ probs,
                                                 const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ seg,
                                                 uint8_t* __restrict__ trimap) {
    const size_t P = (size_t)d.H * d.W, BP = P * d.B;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BP) return;
    const int b = (int)(i / P);
    const int n0 = node_ptr[b], n = node_ptr[b + 1] - n0;
    const int s = seg[i];
    uint8_t lab = GGC_PR_BGD;
    if (s >= 0 && s < n) lab = decide(probs[(size_t)(n0 + s) * 3], probs[(size_t)(n0 + s) * 3 + 2], thr_bg, thr_fg);
    trimap[i] = lab;
}

// ---- S0: _seed_from_prior
__global__ void __launch_bounds__(256) k_seed_flags(TDims d, const uint8_t* __restrict__ trimap, int32_t* __restrict__ flags) {
    const size_t P = (size_t)d.H * d.W;
    const int b = blockIdx.y;
    int f = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t t = trimap[(size_t)b * P + i];
        f |= (t == GGC_FGD || t == GGC_PR_FGD) ? 1 : 2;
    }
    for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o, 64);
    if ((threadIdx.x & 63) == 0 && f) atomicOr(&flags[b], f);
}

// block per image: select the top max(1, round(frac N)) regions of the missing side
__global__ void __launch_bounds__(256) k_seed_select(double seed_frac, const float* __restrict__ prior,
                                                     const int32_t* __restrict__ node_ptr,
                                                     const int32_t* __restrict__ flags, uint8_t* __restrict__ sel) {
    const int b = blockIdx.x;
    const int f = flags[b];
    if (f == 3) return;                      // both sides present: nothing to repair
    const int n0 = node_ptr[b], n = node_ptr[b + 1] - n0;
    int n_seed = (int)rint(seed_frac * (double)n);   // python round(): half to even
    if (n_seed < 1) n_seed = 1;
    const int col = (f & 1) ? 1 : 0;         // FG present -> BG is missing -> column 1
    const float* pr = prior + (size_t)n0 * 3;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float vi = pr[3 * i + col];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float vj = pr[3 * j + col];
            rank += (vj > vi || (vj == vi && j > i)) ? 1 : 0;
        }
        sel[n0 + i] = rank < n_seed ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) k_seed_apply(TDims d, const int32_t* __restrict__ node_ptr,
                                                    const int32_t* __restrict__ flags, const uint8_t* __restrict__ sel,
                                                    const int32_t* __restrict__ seg, uint8_t* __restrict__ trimap) {
    const size_t P = (size_t)d.H * d.W, BP = P * d.B;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BP) return;
    const int b = (int)(i / P);
    const int f = flags[b];
    if (f == 3) return;
    const int n0 = node_ptr[b], n = node_ptr[b + 1] - n0;
    const int s = seg[i];
    if (s >= 0 && s < n && sel[n0 + s]) trimap[i] = (f & 1) ? GGC_PR_BGD : GGC_PR_FGD;
}

// standalone guided filter (pipeline.py:71-100): same box filter and op order as above
__global__ void __launch_bounds__(256) k_gf_prep(size_t BP, const float* __restrict__ guide, const float* __restrict__ src,
                                                 float* __restrict__ planes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BP) return;
    const float g = guide[i], s = src[i];
    planes[i] = g; planes[BP + i] = s; planes[2 * BP + i] = g * s; planes[3 * BP + i] = g * g;
}
__global__ void __launch_bounds__(256) k_gf_ab(size_t BP, float eps, const float* __restrict__ m, float* __restrict__ ab) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BP) return;
    const float mg = m[i], ms = m[BP + i];
    const float cov = m[2 * BP + i] - mg * ms;
    const float var = m[3 * BP + i] - mg * mg;
    const float a = cov / (var + eps);
    ab[i] = a;
    ab[BP + i] = ms - a * mg;
}
__global__ void __launch_bounds__(256) k_gf_final(size_t BP, const float* __restrict__ guide, const float* __restrict__ mab,
                                                  float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < BP) out[i] = mab[i] * guide[i] + mab[BP + i];
}

} // namespace ggc

using namespace ggc;

extern "C" int ggc_refine_trimap(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const float* probs,
                                 const int32_t* node_ptr, const int32_t* segments, const uint8_t* bgr,
                                 float threshold_fg, float threshold_bg, int radius, float eps, int edge_aware,
                                 uint8_t* trimap) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, probs && node_ptr && segments && trimap && (bgr || !edge_aware), GGC_E_INVALID_ARG, "null pointer");
    GGC_REQUIRE(ctx, radius >= 0 && radius <= 256, GGC_E_INVALID_ARG, "filter radius %d out of range", radius);
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const TDims d{B, H, W};
    const size_t BP = (size_t)B * H * W;
    if (!edge_aware) {
        hipLaunchKernelGGL(k_t_plain, dim3(cdiv(BP, 256)), dim3(256), 0, st, d, threshold_fg, threshold_bg, probs,
                           node_ptr, segments, trimap);
        GGC_LAUNCH_CHECK(ctx);
        return GGC_OK;
    }
    float* planes = scratch_t<float>(ctx, S_T_A, BP * 6);
    double* hs = scratch_t<double>(ctx, S_T_B, BP * 6);
    float* means = scratch_t<float>(ctx, S_T_C, BP * 6);
    if (!planes || !hs || !means) return GGC_E_OOM;
    ProfScope prof(ctx, st, "refine_trimap");
    hipLaunchKernelGGL(k_t_prep, dim3(cdiv(BP, 256)), dim3(256), 0, st, d, probs, node_ptr, segments, bgr, planes);
    box_filter(st, d, 6, radius, planes, hs, means);
    float* ab = planes + BP;                      // planes 1..4 are dead now; plane 0 (guide) stays
    hipLaunchKernelGGL(k_t_ab, dim3(cdiv(BP, 256)), dim3(256), 0, st, BP, eps, means, ab);
    box_filter(st, d, 4, radius, ab, hs, means);
    hipLaunchKernelGGL(k_t_final, dim3(cdiv(BP, 256)), dim3(256), 0, st, BP, threshold_fg, threshold_bg, planes, means,
                       trimap);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

extern "C" int ggc_guided_filter(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const float* guide,
                                 const float* src, int radius, float eps, float* out) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, guide && src && out, GGC_E_INVALID_ARG, "null pointer");
    GGC_REQUIRE(ctx, radius >= 0 && radius <= 256, GGC_E_INVALID_ARG, "filter radius %d out of range", radius);
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const TDims d{B, H, W};
    const size_t BP = (size_t)B * H * W;
    float* planes = scratch_t<float>(ctx, S_T_A, BP * 6);
    double* hs = scratch_t<double>(ctx, S_T_B, BP * 6);
    float* means = scratch_t<float>(ctx, S_T_C, BP * 6);
    if (!planes || !hs || !means) return GGC_E_OOM;
    hipLaunchKernelGGL(k_gf_prep, dim3(cdiv(BP, 256)), dim3(256), 0, st, BP, guide, src, planes);
    box_filter(st, d, 4, radius, planes, hs, means);
    float* ab = planes + BP;
    hipLaunchKernelGGL(k_gf_ab, dim3(cdiv(BP, 256)), dim3(256), 0, st, BP, eps, means, ab);
    box_filter(st, d, 2, radius, ab, hs, means);
    hipLaunchKernelGGL(k_gf_final, dim3(cdiv(BP, 256)), dim3(256), 0, st, BP, planes, means, out);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

extern "C" int ggc_seed_from_prior(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const float* prior,
                                   const int32_t* node_ptr, const int32_t* segments, double seed_frac,
                                   uint8_t* trimap) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, prior && node_ptr && segments && trimap, GGC_E_INVALID_ARG, "null pointer");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const TDims d{B, H, W};
    const size_t P = (size_t)H * W, BP = P * B;
    int32_t total_nodes = 0;
    GGC_HIP(ctx, hipMemcpyAsync(&total_nodes, node_ptr + B, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GGC_HIP(ctx, hipStreamSynchronize(st));
    int32_t* flags = scratch_t<int32_t>(ctx, S_MISC_A, (size_t)B);
    uint8_t* sel = scratch_t<uint8_t>(ctx, S_MISC_B, (size_t)std::max(total_nodes, 1));
    if (!flags || !sel) return GGC_E_OOM;
    GGC_HIP(ctx, hipMemsetAsync(flags, 0, sizeof(int32_t) * B, st));
    hipLaunchKernelGGL(k_seed_flags, dim3(std::min(cdiv(P, 256 * 8), 64), B), dim3(256), 0, st, d, trimap, flags);
    hipLaunchKernelGGL(k_seed_select, dim3(B), dim3(256), 0, st, seed_frac, prior, node_ptr, flags, sel);
    hipLaunchKernelGGL(k_seed_apply, dim3(cdiv(BP, 256)), dim3(256), 0, st, d, node_ptr, flags, sel, segments, trimap);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}
