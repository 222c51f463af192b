// ggc_slic.hip — G1: skimage.segmentation.slic as the reference calls it
// (graph_builder.py:177-188; semantics: SURVEY.md Appendix A.1).
//
// Pipeline per batch (all images H x W, one launch per step):
//   k_minmax        global min / max of the Lab input (skimage >= 0.19 rescale)
//   k_lab2          rescale to [0,1] and the second rgb2lab, float32
//   k_gauss<axis>   separable Gaussian, f64 accumulation, y then x, * 1/compactness
//   k_init_centers  regular-grid seeds, colour part 0
//   10 x { k_slic_assign, k_slic_update }
//   k_connectivity  skimage's raster-order connectivity enforcement
//
// Bit-exactness with the CPU path (integer label map) dictates the structure:
//  * assignment is pixel-centric (no atomics): a 32x8 pixel tile gathers, in
//    ascending cluster order, the clusters whose search window meets the tile
//    (wave ballot + prefix compaction into LDS) and every pixel scans that list
//    with a strict '>' — identical to the sequential "lowest k wins ties";
//  * the centroid update is one lane per cluster walking its own window in
//    raster order with float32 running sums — identical rounding to skimage;
//  * all arithmetic is compiled without FMA contraction.
#include "ggc_internal.h"
#include "ggc_math.h"
#include <cmath>
#include <vector>

namespace ggc {

// ---------------------------------------------------------------- min / max
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ void k_minmax_init(int B, uint32_t* mm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}

__global__ void __launch_bounds__(256) k_minmax(size_t n, const float* __restrict__ img, uint32_t* mm) {
    const float* im = img + (size_t)blockIdx.y * n;
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = im[i];
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    // one atomic pair per block: every image has a single (min, max) cell and same-address atomics serialise in L2
    __shared__ float s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
        hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
        atomicMin(&mm[2 * blockIdx.y], f2ord(lo));
        atomicMax(&mm[2 * blockIdx.y + 1], f2ord(hi));
    }
}

// ------------------------------------------------ rescale + second rgb2lab (f32)
__global__ void __launch_bounds__(256) k_lab2(size_t P, const float* __restrict__ img,
                                              const uint32_t* __restrict__ mm, int rescale,
                                              float* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const size_t base = ((size_t)blockIdx.y * P + p) * 3;
    const float mn = ord2f(mm[2 * blockIdx.y]), mx = ord2f(mm[2 * blockIdx.y + 1]);
    const float range = mx - mn;
    float lin[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = img[base + c];
        if (rescale) { v = v - mn; if (mx != mn) v = v / range; }
        lin[c] = v > (float)0.04045 ? (float)det_pow24((double)((v + (float)0.055) / (float)1.055))
                                    : v / (float)12.92;
    }
    const float X = (lin[0] * (float)0.412453 + lin[1] * (float)0.357580) + lin[2] * (float)0.180423;
    const float Y = (lin[0] * (float)0.212671 + lin[1] * (float)0.715160) + lin[2] * (float)0.072169;
    const float Z = (lin[0] * (float)0.019334 + lin[1] * (float)0.119193) + lin[2] * (float)0.950227;
    const float t[3] = {X / (float)0.95047, Y / 1.0f, Z / (float)1.08883};
    float f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        f[c] = t[c] > (float)0.008856 ? (float)det_cbrt((double)t[c]) : (float)7.787 * t[c] + (float)(16.0 / 116.0);
    out[base + 0] = 116.0f * f[1] - 16.0f;
    out[base + 1] = 500.0f * (f[0] - f[1]);
    out[base + 2] = 200.0f * (f[1] - f[2]);
}

// ----------------------------------------------------------------- Gaussian
constexpr int MAX_RADIUS = 16;
struct GaussW { double w[2 * MAX_RADIUS + 1]; int r; };

__device__ __forceinline__ int reflect_sym(int i, int n) { // scipy 'reflect': (b a | a b c | c b)
    if ((unsigned)i < (unsigned)n) return i;              // interior taps skip the run-time modulo
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period; if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// One thread per (pixel, channel); grid = (row segments, rows, images), so no index needs a run-time division.  Same
// op order as scipy's correlate1d for a symmetric kernel: centre tap first, then pairs from the outermost inwards.
// R > 0: radius known at compile time (taps unrolled, every load in flight at once); R = 0: g.r at run time.
template <int AXIS, int R>
__global__ void __launch_bounds__(256) k_gauss(int H, int W, const float* __restrict__ in, GaussW g,
                                               float scale, int apply_scale, float* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;                        // x * 3 + c
    if (col >= 3 * W) return;
    const int y = blockIdx.y;
    const size_t n = (size_t)H * W * 3;
    const float* im = in + (size_t)blockIdx.z * n;
    const int x = col / 3, c = col - 3 * x;
    const int l = AXIS == 0 ? y : x;
    const int len = AXIS == 0 ? H : W;
    auto at = [&](int idx) -> double {
        const int j = reflect_sym(idx, len);
        return (double)(AXIS == 0 ? im[((size_t)j * W + x) * 3 + c] : im[((size_t)y * W + j) * 3 + c]);
    };
    double tmp;
    if (R > 0) {
        constexpr int RR = R > 0 ? R : 1;
        double lo[RR], hi[RR];
        const double mid = at(l);
#pragma unroll
        for (int k = 0; k < R; ++k) { lo[k] = at(l - R + k); hi[k] = at(l + R - k); }
        tmp = mid * g.w[R];
#pragma unroll
        for (int k = 0; k < R; ++k) tmp += (lo[k] + hi[k]) * g.w[k];
    } else {
        tmp = at(l) * g.w[g.r];
        for (int jj = -g.r; jj < 0; ++jj) tmp += (at(l + jj) + at(l - jj)) * g.w[jj + g.r];
    }
    float v = (float)tmp;
    if (apply_scale) v = v * scale;
    out[(size_t)blockIdx.z * n + (size_t)y * W * 3 + col] = v;
}

// The vertical pass with the radius known at compile time: a thread owns one (column, channel) and walks GV_ROWS rows down,
// keeping the 2R+1 taps of the current row in registers and loading ONE new value per row.  Same taps in the same order per
// output as k_gauss<0, R> (centre first, then pairs from the outermost inwards); that kernel's blocks each re-read their
// 2R+1 input rows from memory — 3.3 GB per batch-256 launch for a 0.37 GB image (PMC), 0.43 ms.
constexpr int GV_ROWS = 20;
template <int R>
__global__ void __launch_bounds__(256) k_gauss_v(int H, int W, const float* __restrict__ in, GaussW g, float* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;                        // x * 3 + c
    if (col >= 3 * W) return;
    const size_t n = (size_t)H * W * 3;
    const float* im = in + (size_t)blockIdx.z * n + col;
    float* o = out + (size_t)blockIdx.z * n + col;
    const int y0 = blockIdx.y * GV_ROWS;
    double v[2 * R + 1];                                                   // v[j] = input row y - R + j
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) v[j + 1] = (double)im[(size_t)reflect_sym(y0 - R + j, H) * W * 3];
#pragma unroll 4
    for (int y = y0; y < min(y0 + GV_ROWS, H); ++y) {
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) v[j] = v[j + 1];
        v[2 * R] = (double)im[(size_t)reflect_sym(y + R, H) * W * 3];
        double tmp = v[R] * g.w[R];
#pragma unroll
        for (int k = 0; k < R; ++k) tmp += (v[k] + v[2 * R - k]) * g.w[k];
        o[(size_t)y * W * 3] = (float)tmp;
    }
}

__global__ void __launch_bounds__(256) k_scale(size_t n, const float* __restrict__ in, float scale,
                                               float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * scale;
}

// ------------------------------------------------------------------ k-means
struct SlicGeom {
    int H, W, K;
    int ny, nx, start_y, start_x, step_y, step_x;  // seed grid
    int win_y, win_x;                               // window half-steps from the ACTUAL seed count
    float sw;                                       // spatial weight 1 / step^2
};

// search window of a centre, exactly as _slic_cython truncates it
__device__ __forceinline__ int4 slic_window(float cy, float cx, const SlicGeom& g) {
    if (cy != cy || cx != cx) return make_int4(0, 0, 0, 0);   // dead seed (0/0): never matches again
    float fy0 = cy - (float)(2 * g.win_y); if (!(fy0 > 0.0f)) fy0 = 0.0f;
    float fy1 = cy + (float)(2 * g.win_y) + 1.0f; if (!(fy1 < (float)g.H)) fy1 = (float)g.H;
    float fx0 = cx - (float)(2 * g.win_x); if (!(fx0 > 0.0f)) fx0 = 0.0f;
    float fx1 = cx + (float)(2 * g.win_x) + 1.0f; if (!(fx1 < (float)g.W)) fx1 = (float)g.W;
    return make_int4((int)fy0, (int)fy1, (int)fx0, (int)fx1);  // y0, y1, x0, x1
}

__global__ void k_init_centers(SlicGeom g, float* __restrict__ centers, int4* __restrict__ bounds) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= g.K) return;
    const size_t o = (size_t)blockIdx.y * g.K + k;
    const float cy = (float)(g.start_y + (k / g.nx) * g.step_y);
    const float cx = (float)(g.start_x + (k % g.nx) * g.step_x);
    centers[o * 5 + 0] = cy; centers[o * 5 + 1] = cx;
    centers[o * 5 + 2] = 0.0f; centers[o * 5 + 3] = 0.0f; centers[o * 5 + 4] = 0.0f;
    bounds[o] = slic_window(cy, cx, g);
}

struct Cand { int k, y0, y1, x0, x1; float cy, cx, c0, c1, c2; };

constexpr int TILE_W = 32, TILE_H = 16;     // two pixels per thread: (x, y) and (x, y + 4) inside the wave's 16 x 8 quarter

typedef float v2f __attribute__((ext_vector_type(2)));

// (Round 3 tried pruning: the winner is the lexicographic minimum of (distance, k), so candidates can be scored in any order —
// near ones first, far ones only if their spatial term alone does not exceed every pixel's best.  Same label maps, same
// 0.43 ms per sweep: at compactness 10 the colour term dominates the distance, the spatial bound prunes almost nothing.)
// Pixel-centric assignment.  A block owns a 32 x 16 tile; the clusters whose search window touches the tile are
// compacted (ballot + prefix, ascending k, so the strict '>' keeps skimage's lowest-k tie rule) into LDS once, and
// every thread scores its TWO pixels against each candidate on the packed-f32 pipe (v_pk_*: same IEEE operations in
// the same order as the scalar form, two pixels per instruction, one candidate fetch for both).
__global__ void __launch_bounds__(256) k_slic_assign(SlicGeom g, const float* __restrict__ image,
                                                     const float* __restrict__ centers,
                                                     const int4* __restrict__ bounds,
                                                     int32_t* __restrict__ labels, int32_t* __restrict__ stale) {
    __shared__ Cand cand[256];
    __shared__ int wcount[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int tx0 = blockIdx.x * TILE_W, ty0 = blockIdx.y * TILE_H;
    // a wave owns a compact 16 x 8 quarter of the tile (two pixels per lane, four rows apart): a candidate whose window misses
    // the quarter is skipped by the whole wave (the exec-zero branch below) — 24 of the tile's ~33 candidates reach a quarter,
    // where every wave used to walk all of them for its two full-width row pairs
    const int px = tx0 + (wave & 1) * 16 + (lane & 15), pya = ty0 + (wave >> 1) * 8 + (lane >> 4), pyb = pya + 4;
    const bool ina = px < g.W && pya < g.H, inb = px < g.W && pyb < g.H;
    const size_t P = (size_t)g.H * g.W;
    const size_t pa = (size_t)b * P + (size_t)pya * g.W + px, pb = pa + (size_t)4 * g.W;
    v2f i0 = 0.f, i1 = 0.f, i2 = 0.f;
    if (ina) { i0.x = image[3 * pa]; i1.x = image[3 * pa + 1]; i2.x = image[3 * pa + 2]; }
    if (inb) { i0.y = image[3 * pb]; i1.y = image[3 * pb + 1]; i2.y = image[3 * pb + 2]; }
    const v2f fy = {(float)pya, (float)pyb};
    const float fx = (float)px;
    float best_a = INFINITY, best_b = INFINITY;
    int lab_a = -1, lab_b = -1;
    const float* cen = centers + (size_t)b * g.K * 5;
    const int4* bnd = bounds + (size_t)b * g.K;
    for (int base = 0; base < g.K; base += 256) {
        const int k = base + tid;
        bool hit = false;
        int4 bd = make_int4(0, 0, 0, 0);
        if (k < g.K) {
            bd = bnd[k];
            hit = bd.x < ty0 + TILE_H && bd.y > ty0 && bd.z < tx0 + TILE_W && bd.w > tx0 && bd.y > bd.x && bd.w > bd.z;
        }
        const unsigned long long bal = __ballot(hit);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) off += wcount[w]; total += wcount[w]; }
        if (hit) {
            Cand c;
            c.k = k; c.y0 = bd.x; c.y1 = bd.y; c.x0 = bd.z; c.x1 = bd.w;
            c.cy = cen[k * 5 + 0]; c.cx = cen[k * 5 + 1];
            c.c0 = cen[k * 5 + 2]; c.c1 = cen[k * 5 + 3]; c.c2 = cen[k * 5 + 4];
            cand[off + pre] = c;
        }
        __syncthreads();
        for (int i = 0; i < total; ++i) {
            const Cand& c = cand[i];
            const bool xin = (unsigned)(px - c.x0) < (unsigned)(c.x1 - c.x0);
            const unsigned hgt = (unsigned)(c.y1 - c.y0);
            const bool hit_a = xin && (unsigned)(pya - c.y0) < hgt, hit_b = xin && (unsigned)(pyb - c.y0) < hgt;
            if (!(hit_a || hit_b)) continue;
            const v2f ty = c.cy - fy;
            const float tx = c.cx - fx;
            v2f dd = (ty * ty + tx * tx) * g.sw;
            v2f t = i0 - c.c0;
            v2f dc = t * t;                     // (0 + t*t) == t*t exactly
            t = i1 - c.c1; dc += t * t;
            t = i2 - c.c2; dc += t * t;
            dd += dc;
            if (hit_a && best_a > dd.x) { best_a = dd.x; lab_a = c.k; }
            if (hit_b && best_b > dd.y) { best_b = dd.y; lab_b = c.k; }
        }
        __syncthreads();
    }
    bool none = false;
    if (ina) { if (lab_a >= 0) labels[pa] = lab_a; else none = true; }
    if (inb) { if (lab_b >= 0) labels[pb] = lab_b; else none = true; }
    if (none) atomicOr(&stale[b], 1);   // no window covers the pixel: it keeps its previous label
}

// Centre update.  skimage accumulates float32 running sums over a cluster's pixels in raster order, so the
// additions of the three colour channels must stay sequential per cluster — but nothing else has to:
//  * a group of 8 lanes owns a cluster (8 clusters per wave, neighbours along the seed grid, so their windows
//    and shapes are alike; 16 and 4 lanes measured slower): the group reads a window row as chunks of 8 consecutive
//    labels, the loads of a whole row block — and those of the next one — in flight together (the scan is otherwise
//    a chain of L2 round trips), a ballot marks the cluster's pixels, the matching lanes fetch their colours;
//  * the coordinate sums are sums of small integers: exact in float32 in any order while they stay below 2^24, so
//    they are taken as integer popcount / lane-local sums (a cluster that large falls back to the ordered loop);
//  * the colour fold walks the set bits of each group's ballot in ascending x with one ds_bpermute per channel,
//    all groups in the same wave instruction.  Same additions, same order, a fraction of the instructions.
// Measured in round 3 and not kept (same label maps each time): (a) a MEMBER BOX per cluster, grown by k_slic_assign with
// integer min / max (ballots per wave, a 64-slot LDS table per block, 4 global atomics per cluster and block) and scanned
// here instead of the (4 step + 1)^2 window — this kernel 0.58 -> 0.44 ms per sweep, but the assignment 0.42 -> 0.60: the
// scan is not what bounds the update, the ordered fold is; (b) one LANE per cluster walking its member box with plain
// predicated adds — 0.60 ms per sweep: 64 clusters per wave-load are 32-64 cache lines.
constexpr int UPD_LANES = 8, UPD_GROUPS = 64 / UPD_LANES, UPD_CH = 8;

__global__ void __launch_bounds__(256) k_slic_update(SlicGeom g, const float* __restrict__ image,
                                                     const int32_t* __restrict__ labels,
                                                     const int32_t* __restrict__ stale,
                                                     float* __restrict__ centers, int4* __restrict__ bounds) {
    const int lane = threadIdx.x & 63, sub = lane / UPD_LANES, sl = lane % UPD_LANES;
    const int k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * UPD_GROUPS + sub;
    const bool live = k < g.K;
    const int b = blockIdx.y;
    const size_t o = (size_t)b * g.K + (live ? k : 0);
    int4 bd = live ? bounds[o] : make_int4(0, 0, 0, 0);
    if (live && stale[b]) bd = make_int4(0, g.H, 0, g.W);   // some pixel kept an old label: scan everything
    const size_t P = (size_t)g.H * g.W;
    const int32_t* lb = labels + (size_t)b * P;
    const float* im = image + (size_t)b * P * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    int cnt = 0, isy = 0, isx = 0;                          // isx: this lane's share, reduced at the end
    const int rows = bd.y - bd.x, chunks = (bd.w - bd.z + UPD_LANES - 1) / UPD_LANES;
    int rows_max = rows, chunks_max = chunks;               // the wave runs the longest of its four loops
#pragma unroll
    for (int off = UPD_LANES; off < 64; off <<= 1) {
        rows_max = max(rows_max, __shfl_xor(rows_max, off, 64));
        chunks_max = max(chunks_max, __shfl_xor(chunks_max, off, 64));
    }
    // labels of (row ry, chunks c0 .. c0 + UPD_CH - 1); -1 outside the group's window.  One pointer per lane and row, the
    // chunks at constant offsets from it, the lane's number of chunks inside the window counted once (the kernel is bound by
    // instruction issue: the per-chunk 64-bit index arithmetic was a third of the row's instructions)
    const int32_t* lane_p = lb + (size_t)bd.x * g.W + bd.z + sl;
    const int nvc = (bd.w - bd.z - sl + UPD_LANES - 1) / UPD_LANES;           // chunks c with bd.z + c * UPD_LANES + sl < bd.w
    auto load_row = [&](int ry, int c0, int (&lab)[UPD_CH]) {
        const int32_t* rp = lane_p + (size_t)ry * g.W + c0 * UPD_LANES;
        const int nv = ry < rows ? nvc - c0 : 0;
#pragma unroll
        for (int c = 0; c < UPD_CH; ++c) lab[c] = c < nv ? rp[c * UPD_LANES] : -1;
    };
    // blocks of UPD_CH chunks in raster order: (row 0, block 0), (row 0, block 1), ..., (row 1, block 0), ...
    const int nblk = (chunks_max + UPD_CH - 1) / UPD_CH, total = rows_max * nblk;
    int cur[UPD_CH], nxt[UPD_CH];
    if (total > 0) load_row(0, 0, cur);
    int ry = 0, cb = 0;
    for (int t = 0; t < total; ++t) {
        int ry_n = ry, cb_n = cb + 1;
        if (cb_n == nblk) { cb_n = 0; ++ry_n; }
        load_row(ry_n, cb_n * UPD_CH, nxt);                 // next block in flight while this one is folded (row rows_max reads nothing)
        const int y = bd.x + ry;
#pragma unroll
        for (int c = 0; c < UPD_CH; ++c) {
            const bool hit = cur[c] == k && live;
            const unsigned long long bal = __ballot(hit);
            if (bal) {
                const int x = bd.z + (cb * UPD_CH + c) * UPD_LANES + sl;
                unsigned int m = (unsigned int)(bal >> (sub * UPD_LANES)) & ((1u << UPD_LANES) - 1u);
                float v0 = 0.f, v1 = 0.f, v2 = 0.f;
                if (hit) { const float* px = im + ((size_t)y * g.W + x) * 3; v0 = px[0]; v1 = px[1]; v2 = px[2]; isx += x; }
                const int n = __popc(m);
                cnt += n; isy += n * y;
                // the chunk's 8 pixels in x order, every lane of the group keeping the same running sums: lane u's value is
                // broadcast by ds_swizzle (no address register) and added unconditionally — a lane without a hit holds +0,
                // and s + (+0) == s bit for bit (s never is -0: it starts at +0).  The loop over the set bits this replaces
                // spent 14 instructions per member pixel on find-first-set, three ds_bpermute and the loop test.
                static_assert(UPD_LANES == 8, "the broadcast pattern is written for groups of 8 lanes");
#define GGC_UPD_STEP(U) { s0 += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v0), 0x18 | ((U) << 5))); \
                          s1 += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v1), 0x18 | ((U) << 5))); \
                          s2 += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v2), 0x18 | ((U) << 5))); }
                GGC_UPD_STEP(0) GGC_UPD_STEP(1) GGC_UPD_STEP(2) GGC_UPD_STEP(3) GGC_UPD_STEP(4) GGC_UPD_STEP(5) GGC_UPD_STEP(6) GGC_UPD_STEP(7)
#undef GGC_UPD_STEP
            }
        }
#pragma unroll
        for (int c = 0; c < UPD_CH; ++c) cur[c] = nxt[c];
        ry = ry_n; cb = cb_n;
    }
#pragma unroll
    for (int off = 1; off < UPD_LANES; off <<= 1) isx += __shfl_xor(isx, off, 64);
    float sy = (float)isy, sx = (float)isx;
    if (live && (isy >= (1 << 24) || isx >= (1 << 24))) {
        // (never with SLIC-sized clusters) float32 running sums stop being exact: redo them in order
        sy = 0.f; sx = 0.f;
        for (int y = bd.x; y < bd.y; ++y)
            for (int x = bd.z; x < bd.w; ++x)
                if (lb[(size_t)y * g.W + x] == k) { sy += (float)y; sx += (float)x; }
    }
    if (!live || sl != 0) return;
    const float n = (float)cnt;
    const float cy = sy / n, cx = sx / n;
    centers[o * 5 + 0] = cy; centers[o * 5 + 1] = cx;
    centers[o * 5 + 2] = s0 / n; centers[o * 5 + 3] = s1 / n; centers[o * 5 + 4] = s2 / n;
    bounds[o] = slic_window(cy, cx, g);
}

// ------------------------------------------------ connectivity enforcement
// skimage's _enforce_label_connectivity_cython is defined by a raster scan whose
// relabelling depends on visit order (SURVEY hard part 1.iii).  This version
// replays it literally, one thread per image (images of a batch in parallel).
__global__ void __launch_bounds__(64) k_connectivity_seq(int H, int W, const int32_t* __restrict__ labels,
                                                         int min_size, int max_size,
                                                         int32_t* __restrict__ queue,
                                                         int32_t* __restrict__ out, int32_t* __restrict__ n_nodes) {
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    const size_t P = (size_t)H * W;
    const int32_t* lb = labels + (size_t)b * P;
    int32_t* o = out + (size_t)b * P;
    int32_t* q = queue + (size_t)b * (size_t)(max_size > 0 ? max_size : 1);
    int cur = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int p0 = y * W + x;
            if (o[p0] >= 0) continue;
            int adjacent = 0;
            const int label = lb[p0];
            o[p0] = cur;
            int size = 1, visited = 0;
            q[0] = p0;
            while (visited < size && size < max_size) {
                const int pc = q[visited];
                const int yc = pc / W, xc = pc - yc * W;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int xx = xc + (i == 0 ? 1 : i == 1 ? -1 : 0);
                    const int yy = yc + (i == 2 ? 1 : i == 3 ? -1 : 0);
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    const int pp = yy * W + xx;
                    const int op = o[pp];
                    if (op == -1 && lb[pp] == label) {
                        o[pp] = cur;
                        q[size] = pp;
                        size += 1;
                        if (size >= max_size) break;
                    } else if (op >= 0 && op != cur) {
                        adjacent = op;
                    }
                }
                visited += 1;
            }
            if (size < min_size) {
                for (int i = 0; i < size; ++i) o[q[i]] = adjacent;
            } else {
                cur += 1;
            }
        }
    n_nodes[b] = cur > 0 ? cur : 1;
}

// ------------------------------------------- connectivity enforcement, parallel
// Same result as the raster scan above, restructured so that only the
// order-dependent parts stay sequential (SURVEY hard part 1.iii):
//   1. union-find CCL of the raw label map (4-connectivity); a component's root
//      is its smallest pixel index = the pixel where the raster scan discovers it;
//   2. components of >= max_size pixels: one lane replays the capped BFS from the
//      root (exactly max_size pixels in queue order), the carved piece keeps the
//      root, the remainder is re-labelled by another CCL round (host loop; real
//      SLIC output needs zero or one extra round);
//   3. components of >= min_size pixels keep their discovery rank (prefix sum over
//      the roots in raster order) as their label;
//   4. components of < min_size pixels: one lane replays the BFS and remembers the
//      last neighbour pixel that belongs to an earlier-discovered component; the
//      label is inherited along that pointer chain (0 when there is none).
// Validated pixel-for-pixel against the sequential kernel and the CPU oracle.
__device__ __forceinline__ int cn_find(const int32_t* parent, int i) {
    int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) { i = p; p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return i;
}
__device__ __forceinline__ void cn_union(int32_t* parent, int a, int b) {
    for (;;) {
        a = cn_find(parent, a); b = cn_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

struct CnDims { int B, H, W, P, min_size, max_size; };

// Horizontal runs need no atomics: inside a wave's 64 consecutive pixels a pixel's parent is set straight to the first
// pixel of its run (ballot of the run breaks), which is also the run's smallest index.  k_cn_merge then only joins
// runs: across a 64-pixel boundary, and vertically where a run first touches a run of the row above.
__global__ void __launch_bounds__(256) k_cn_reset(CnDims d, const int32_t* __restrict__ raw, const uint8_t* __restrict__ pending,
                                                  int32_t* __restrict__ parent, int32_t* __restrict__ csize) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < (size_t)d.B * d.P && pending[i];
    bool same_left = false;
    int p = 0;
    if (valid) {
        p = (int)(i % d.P);
        same_left = (p % d.W) > 0 && pending[i - 1] && raw[i - 1] == raw[i];
    }
    const unsigned long long starts = __ballot(valid && (!same_left || lane == 0));
    if (!valid) return;
    const int start_lane = 63 - __clzll((long long)(starts & ((2ull << lane) - 1ull)));
    parent[i] = p - (lane - start_lane);
    csize[i] = 0;
}

__global__ void __launch_bounds__(256) k_cn_merge(CnDims d, const int32_t* __restrict__ raw, const uint8_t* __restrict__ pending,
                                                  int32_t* __restrict__ parent) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= d.W || y >= d.H) return;
    const size_t base = (size_t)blockIdx.z * d.P;
    const int p = y * d.W + x;
    if (!pending[base + p]) return;
    const int l = raw[base + p];
    const bool same_left = x > 0 && pending[base + p - 1] && raw[base + p - 1] == l;
    if (same_left && ((base + p) & 63) == 0) cn_union(parent + base, p, p - 1);      // run continues across k_cn_reset's wave boundary
    if (y > 0 && pending[base + p - d.W] && raw[base + p - d.W] == l) {
        // the left neighbour already joined this pair of runs if it sits under the same upper run
        const bool joined = same_left && pending[base + p - d.W - 1] && raw[base + p - d.W - 1] == l;
        if (!joined) cn_union(parent + base, p, p - d.W);
    }
}

__global__ void __launch_bounds__(256) k_cn_size(CnDims d, const int32_t* __restrict__ raw, const uint8_t* __restrict__ pending,
                                                 int32_t* __restrict__ parent, int32_t* __restrict__ csize) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < (size_t)d.B * d.P && pending[i];
    bool same_left = false;
    size_t base = 0;
    int r = 0;
    if (valid) {
        base = (i / d.P) * d.P;
        const int p = (int)(i - base);
        same_left = (p % d.W) > 0 && pending[i - 1] && raw[i - 1] == raw[i];
        r = cn_find(parent + base, p);
        parent[i] = r;
    }
    // one atomic per run (the runs of k_cn_reset), not per pixel
    const unsigned long long vmask = __ballot(valid);
    const unsigned long long starts = __ballot(valid && (!same_left || lane == 0));
    if (valid && ((starts >> lane) & 1ull)) {
        const unsigned long long stop = (starts | ~vmask) >> lane >> 1;       // next run start or hole above this lane
        const int len = stop ? __ffsll((long long)stop) : 64 - lane;
        atomicAdd(&csize[base + r], len);
    }
}

// components below max_size are final; count the ones that still need carving
__global__ void __launch_bounds__(256) k_cn_settle(CnDims d, uint8_t* __restrict__ pending, const int32_t* __restrict__ parent,
                                                   const int32_t* __restrict__ csize, int32_t* __restrict__ n_big) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P || !pending[i]) return;
    const size_t base = (i / d.P) * d.P;
    const int p = (int)(i - base), r = parent[i];
    if (csize[base + r] < d.max_size) pending[i] = 0;
    else if (p == r) atomicAdd(n_big, 1);
}

// replay of the capped BFS for every root of a too-large component (one lane each)
__global__ void __launch_bounds__(64) k_cn_carve(CnDims d, int tag, uint8_t* __restrict__ pending, const int32_t* __restrict__ parent,
                                                 int32_t* __restrict__ csize, int32_t* __restrict__ mark,
                                                 int32_t* __restrict__ queue, int32_t* __restrict__ qtop) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    const int b = (int)(i / d.P);
    const size_t base = (size_t)b * d.P;
    const int r = (int)(i - base);
    if (!pending[i] || parent[i] != r || csize[i] < d.max_size) return;
    int32_t* q = queue + base + atomicAdd(&qtop[b], d.max_size);
    int size = 1, visited = 0;
    q[0] = r; mark[base + r] = tag;
    while (visited < size && size < d.max_size) {
        const int pc = q[visited];
        const int yc = pc / d.W, xc = pc - yc * d.W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = xc + (k == 0 ? 1 : k == 1 ? -1 : 0), yy = yc + (k == 2 ? 1 : k == 3 ? -1 : 0);
            if (xx < 0 || xx >= d.W || yy < 0 || yy >= d.H) continue;
            const int pp = yy * d.W + xx;
            if (pending[base + pp] && parent[base + pp] == r && mark[base + pp] != tag) {
                mark[base + pp] = tag;
                q[size++] = pp;
                if (size >= d.max_size) break;
            }
        }
        visited += 1;
    }
    for (int k = 0; k < size; ++k) pending[base + q[k]] = 0;   // the carved piece is final, root r
    csize[i] = size;
}

// kept flag at the roots (1 = keeps its own label), everything else 0; scanned per image afterwards
__global__ void __launch_bounds__(256) k_cn_kept(CnDims d, const int32_t* __restrict__ parent, const int32_t* __restrict__ csize,
                                                 int32_t* __restrict__ rank) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    rank[i] = (parent[i] == (int)(i % d.P) && csize[i] >= d.min_size) ? 1 : 0;
}

// per-image exclusive scan (block per image); n_nodes = max(total, 1).  The block walks the image in tiles of 4096
// elements — four consecutive ones per thread, so a wave reads 1 KB at a stretch — scanning each tile through wave shuffles
// and carrying the running total (a thread-per-chunk scan read 118 strided words per thread: 558 us per batch of 256).
__global__ void __launch_bounds__(1024) k_cn_scan(CnDims d, int32_t* __restrict__ rank, int32_t* __restrict__ n_nodes) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t s_carry;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* a = rank + (size_t)b * d.P;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < d.P; base += 4096) {
        const int i0 = base + tid * 4;
        int32_t c[4], s = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = i0 + j < d.P ? a[i0 + j] : 0; s += c[j]; }
        int32_t incl = s;
        for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int32_t run = s_carry + (incl - s), total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { if (w < wave) run += wsum[w]; total += wsum[w]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (i0 + j < d.P) a[i0 + j] = run; run += c[j]; }
        __syncthreads();
        if (tid == 0) s_carry += total;
        __syncthreads();
    }
    if (tid == 0) n_nodes[b] = s_carry > 0 ? s_carry : 1;
}

// small components: replay the BFS, remember the last neighbour of an earlier component
__global__ void __launch_bounds__(64) k_cn_small(CnDims d, int tag, const int32_t* __restrict__ parent,
                                                 const int32_t* __restrict__ csize, int32_t* __restrict__ mark,
                                                 int32_t* __restrict__ queue, int32_t* __restrict__ qtop,
                                                 int32_t* __restrict__ tgt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    const int b = (int)(i / d.P);
    const size_t base = (size_t)b * d.P;
    const int r = (int)(i - base);
    if (parent[i] != r || csize[i] >= d.min_size) return;
    const int n = csize[i];
    int32_t* q = queue + base + atomicAdd(&qtop[b], n);
    int size = 1, visited = 0, last = -1;
    q[0] = r; mark[base + r] = tag;
    while (visited < size) {
        const int pc = q[visited];
        const int yc = pc / d.W, xc = pc - yc * d.W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = xc + (k == 0 ? 1 : k == 1 ? -1 : 0), yy = yc + (k == 2 ? 1 : k == 3 ? -1 : 0);
            if (xx < 0 || xx >= d.W || yy < 0 || yy >= d.H) continue;
            const int pp = yy * d.W + xx;
            const int rp = parent[base + pp];
            if (rp == r) { if (mark[base + pp] != tag && size < n) { mark[base + pp] = tag; q[size++] = pp; } }
            else if (rp < r) last = rp;      // discovered earlier => already labelled when the scan gets here
        }
        visited += 1;
    }
    tgt[i] = last;
}

__global__ void __launch_bounds__(256) k_cn_write(CnDims d, const int32_t* __restrict__ parent, const int32_t* __restrict__ csize,
                                                  const int32_t* __restrict__ rank, const int32_t* __restrict__ tgt,
                                                  int32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    const size_t base = (i / d.P) * d.P;
    int r = parent[i];
    int lab = 0;
    for (;;) {
        if (csize[base + r] >= d.min_size) { lab = rank[base + r]; break; }
        const int t = tgt[base + r];
        if (t < 0) { lab = 0; break; }
        r = t;
    }
    out[i] = lab;
}

// ------------------------------------------------------------- host helpers

// numpy's pairwise summation for n <= 128 (scipy normalises the kernel with ndarray.sum())
static double np_pairwise_sum(const double* a, int n) {
    if (n < 8) { double s = 0.0; for (int i = 0; i < n; ++i) s += a[i]; return s; }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// skimage.util.regular_grid((1, H, W), n) (util/_regular_grid.py:61-83)
struct Grid { int step_y, step_x, start_y, start_x, ny, nx, K; };
static Grid regular_grid(int H, int W, int n_points) {
    Grid g{};
    const double dims[3] = {1.0, (double)(H <= W ? H : W), (double)(H <= W ? W : H)};
    const double space = 1.0 * H * W;
    if (space <= (double)n_points) { g = {1, 1, 0, 0, H, W, H * W}; return g; }
    double st[3];
    for (int i = 0; i < 3; ++i) st[i] = std::pow(space / n_points, 1.0 / 3.0);
    if (dims[0] < st[0] || dims[1] < st[1] || dims[2] < st[2]) {
        for (int dim = 0; dim < 3; ++dim) {
            st[dim] = dims[dim];
            double sp = 1.0;
            for (int j = dim + 1; j < 3; ++j) sp *= dims[j];
            for (int j = dim + 1; j < 3; ++j) st[j] = std::pow(sp / n_points, 1.0 / (3 - dim - 1));
            if (dims[0] >= st[0] && dims[1] >= st[1] && dims[2] >= st[2]) break;
        }
    }
    int starts[3], steps[3];
    for (int i = 0; i < 3; ++i) { starts[i] = (int)std::floor(st[i] / 2.0); steps[i] = (int)std::rint(st[i]); }
    const int iy = (H <= W) ? 1 : 2, ix = (H <= W) ? 2 : 1;
    g.step_y = steps[iy]; g.start_y = starts[iy];
    g.step_x = steps[ix]; g.start_x = starts[ix];
    g.ny = std::max(0, (H - g.start_y + g.step_y - 1) / g.step_y);
    g.nx = std::max(0, (W - g.start_x + g.step_x - 1) / g.step_x);
    g.K = g.ny * g.nx;
    return g;
}

} // namespace ggc

using namespace ggc;

static int enforce_connectivity(ggc_ctx* ctx, hipStream_t st, int B, int H, int W, const int32_t* raw, int min_size,
                                int max_size, int32_t* segments, int32_t* n_nodes) {
    const size_t P = (size_t)H * W;
    if (knobs().slic_seq_connectivity) {   // literal raster-scan replay, one thread per image (A/B reference)
        int32_t* queue = scratch_t<int32_t>(ctx, S_SLIC_AUX3, (size_t)B * (size_t)std::max(max_size, 1));
        if (!queue) return GGC_E_OOM;
        GGC_HIP(ctx, hipMemsetAsync(segments, 0xFF, sizeof(int32_t) * (size_t)B * P, st));
        ProfScope prof(ctx, st, "slic_connectivity");
        hipLaunchKernelGGL(k_connectivity_seq, dim3(B), dim3(64), 0, st, H, W, raw, min_size, max_size, queue, segments,
                           n_nodes);
        GGC_LAUNCH_CHECK(ctx);
        return GGC_OK;
    }
    {
        ProfScope prof(ctx, st, "slic_connectivity");
        const size_t BP = (size_t)B * P;
        const CnDims cd{B, H, W, (int)P, min_size, std::max(max_size, 1)};
        int32_t* work = scratch_t<int32_t>(ctx, S_SLIC_AUX3, BP * 6 + (size_t)B * (size_t)cd.max_size + 2 * (size_t)B + 16);
        uint8_t* pending = scratch_t<uint8_t>(ctx, S_SLIC_AUX4, BP);
        if (!work || !pending) return GGC_E_OOM;
        int32_t *parent = work, *csize = work + BP, *rank = work + 2 * BP, *tgt = work + 3 * BP, *mark = work + 4 * BP;
        int32_t* queue = work + 5 * BP;                       // per image P (+ max_size slack at the very end)
        int32_t* qtop = work + 6 * BP + (size_t)B * cd.max_size;
        int32_t* n_big = qtop + B;
        GGC_HIP(ctx, hipMemsetAsync(pending, 1, BP, st));
        GGC_HIP(ctx, hipMemsetAsync(mark, 0, sizeof(int32_t) * BP, st));
        const dim3 g1(cdiv(BP, 256)), g2(cdiv(W, 64), cdiv(H, 4), B), g64(cdiv(BP, 64));
        for (int round = 0; round < 4096; ++round) {
            GGC_HIP(ctx, hipMemsetAsync(qtop, 0, sizeof(int32_t) * (B + 1), st));
            hipLaunchKernelGGL(k_cn_reset, g1, dim3(256), 0, st, cd, raw, pending, parent, csize);
            hipLaunchKernelGGL(k_cn_merge, g2, dim3(256), 0, st, cd, raw, pending, parent);
            hipLaunchKernelGGL(k_cn_size, g1, dim3(256), 0, st, cd, raw, pending, parent, csize);
            hipLaunchKernelGGL(k_cn_settle, g1, dim3(256), 0, st, cd, pending, parent, csize, n_big);
            GGC_LAUNCH_CHECK(ctx);
            int32_t h_big = 0;
            GGC_HIP(ctx, hipMemcpyAsync(&h_big, n_big, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            GGC_HIP(ctx, hipStreamSynchronize(st));
            if (h_big == 0) break;
            hipLaunchKernelGGL(k_cn_carve, g64, dim3(64), 0, st, cd, round + 1, pending, parent, csize, mark, queue, qtop);
            GGC_LAUNCH_CHECK(ctx);
        }
        hipLaunchKernelGGL(k_cn_kept, g1, dim3(256), 0, st, cd, parent, csize, rank);
        hipLaunchKernelGGL(k_cn_scan, dim3(B), dim3(1024), 0, st, cd, rank, n_nodes);
        GGC_HIP(ctx, hipMemsetAsync(qtop, 0, sizeof(int32_t) * (B + 1), st));
        hipLaunchKernelGGL(k_cn_small, g64, dim3(64), 0, st, cd, 0x7FFFFFFF, parent, csize, mark, queue, qtop, tgt);
        hipLaunchKernelGGL(k_cn_write, g1, dim3(256), 0, st, cd, parent, csize, rank, tgt, segments);
        GGC_LAUNCH_CHECK(ctx);
    }
    return GGC_OK;
}

extern "C" int ggc_slic(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const float* image,
                        int n_segments, float compactness, float sigma, int rescale_input,
                        int32_t* segments, int32_t* n_nodes) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, image && segments && n_nodes, GGC_E_INVALID_ARG, "null pointer");
    GGC_REQUIRE(ctx, n_segments >= 1 && compactness > 0.0f && sigma >= 0.0f, GGC_E_INVALID_ARG,
                "bad SLIC parameters n_segments=%d compactness=%g sigma=%g", n_segments, compactness, sigma);
    GGC_REQUIRE(ctx, (size_t)H * W < (1u << 30), GGC_E_SHAPE, "image too large");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t P = (size_t)H * W;

    const Grid seed = regular_grid(H, W, n_segments);
    GGC_REQUIRE(ctx, seed.K >= 1, GGC_E_SHAPE, "no SLIC seeds for %dx%d with n_segments=%d", H, W, n_segments);
    const Grid win = regular_grid(H, W, seed.K);   // _slic_cython recomputes the grid from the seed count
    SlicGeom g{};
    g.H = H; g.W = W; g.K = seed.K;
    g.ny = seed.ny; g.nx = seed.nx; g.start_y = seed.start_y; g.start_x = seed.start_x;
    g.step_y = seed.step_y; g.step_x = seed.step_x;
    g.win_y = win.step_y; g.win_x = win.step_x;
    const float step = (float)std::max(std::max(seed.step_y, seed.step_x), 1);
    g.sw = (float)(1.0 / (double)(step * step));

    float* img_a = scratch_t<float>(ctx, S_SLIC_IMG, (size_t)B * P * 3);
    float* img_b = scratch_t<float>(ctx, S_SLIC_TMP, (size_t)B * P * 3);
    uint32_t* mm = scratch_t<uint32_t>(ctx, S_SLIC_MINMAX, (size_t)B * 2);
    float* centers = scratch_t<float>(ctx, S_SLIC_CENTERS, (size_t)B * g.K * 5);
    int4* bounds = scratch_t<int4>(ctx, S_SLIC_AUX, (size_t)B * g.K);
    int32_t* raw = scratch_t<int32_t>(ctx, S_SLIC_LABELS, (size_t)B * P);
    int32_t* stale = scratch_t<int32_t>(ctx, S_SLIC_AUX2, (size_t)B * 10);
    if (!img_a || !img_b || !mm || !centers || !bounds || !raw || !stale) return GGC_E_OOM;

    // 1-2: rescale + second Lab
    hipLaunchKernelGGL(k_minmax_init, dim3(cdiv(B, 256)), dim3(256), 0, st, B, mm);
    hipLaunchKernelGGL(k_minmax, dim3(std::min(cdiv(P * 3, 256 * 8), 48), B), dim3(256), 0, st, P * 3, image, mm);
    hipLaunchKernelGGL(k_lab2, dim3(cdiv(P, 256), B), dim3(256), 0, st, P, image, mm, rescale_input, img_a);
    GGC_LAUNCH_CHECK(ctx);
    // 4-5: Gaussian (y then x) and the 1/compactness scale
    const float ratio = (float)(1.0 / (double)compactness);
    const float* km_img = nullptr;
    if (sigma > 0.0f) {
        GaussW gw{};
        gw.r = (int)(4.0 * (double)sigma + 0.5);
        GGC_REQUIRE(ctx, gw.r <= MAX_RADIUS, GGC_E_UNSUPPORTED, "sigma=%g needs radius %d > %d", sigma, gw.r, MAX_RADIUS);
        ggc_gaussian_taps((double)sigma, gw.r, gw.w);                      // numpy's exp values at sigma = 1 (include/ggc_fmath.h)
        const double sum = np_pairwise_sum(gw.w, 2 * gw.r + 1);
        for (int i = 0; i < 2 * gw.r + 1; ++i) gw.w[i] = gw.w[i] / sum;
        GGC_REQUIRE(ctx, H <= 65535, GGC_E_SHAPE, "H=%d exceeds the launch grid", H);
        const dim3 grid(cdiv((size_t)W * 3, 256), H, B);
        if (gw.r == 4) {                                                   // sigma = 1, the pipeline's setting
            hipLaunchKernelGGL((k_gauss_v<4>), dim3(cdiv((size_t)W * 3, 256), cdiv(H, GV_ROWS), B), dim3(256), 0, st, H, W, img_a, gw, img_b);
            hipLaunchKernelGGL((k_gauss<1, 4>), grid, dim3(256), 0, st, H, W, img_b, gw, ratio, 1, img_a);
        } else {
            hipLaunchKernelGGL((k_gauss<0, 0>), grid, dim3(256), 0, st, H, W, img_a, gw, 1.0f, 0, img_b);
            hipLaunchKernelGGL((k_gauss<1, 0>), grid, dim3(256), 0, st, H, W, img_b, gw, ratio, 1, img_a);
        }
        km_img = img_a;
    } else {
        hipLaunchKernelGGL(k_scale, dim3(cdiv((size_t)B * P * 3, 256)), dim3(256), 0, st, (size_t)B * P * 3, img_a,
                           ratio, img_b);
        km_img = img_b;
    }
    GGC_LAUNCH_CHECK(ctx);
    // 6: k-means, exactly 10 sweeps (the early exit of _slic_cython can never trigger)
    hipLaunchKernelGGL(k_init_centers, dim3(cdiv(g.K, 256), B), dim3(256), 0, st, g, centers, bounds);
    GGC_HIP(ctx, hipMemsetAsync(raw, 0, sizeof(int32_t) * (size_t)B * P, st));
    GGC_HIP(ctx, hipMemsetAsync(stale, 0, sizeof(int32_t) * (size_t)B * 10, st));
    for (int it = 0; it < 10; ++it) {
        {
            ProfScope prof(ctx, st, "slic_assign");
            hipLaunchKernelGGL(k_slic_assign, dim3(cdiv(W, TILE_W), cdiv(H, TILE_H), B), dim3(256), 0, st, g, km_img,
                               centers, bounds, raw, stale + (size_t)it * B);
        }
        {
            ProfScope prof(ctx, st, "slic_update");
            hipLaunchKernelGGL(k_slic_update, dim3(cdiv(g.K, 4 * UPD_GROUPS), B), dim3(256), 0, st, g, km_img, raw,
                               stale + (size_t)it * B, centers, bounds);
        }
        GGC_LAUNCH_CHECK(ctx);
    }
    // 7: connectivity
    const double seg_size = (double)P / (double)g.K;
    const int min_size = (int)(0.5 * seg_size), max_size = (int)(3.0 * seg_size);
    return enforce_connectivity(ctx, st, B, H, W, raw, min_size, max_size, segments, n_nodes);
}

// =====================================================================================================================
// use_lab=False: the reference hands `self.rgb.astype(float)` to slic (graph_builder.py:177-179).  A float64 input keeps every
// stage of skimage's slic in float64 (rgb2lab, the Gaussian — including its pass along the depth-1 axis — and the double
// instance of _slic_cython), so this is the path above once more in double.  It is the reference's NON-default option:
// the kernels are plain (no packed math, one cluster per 8-lane group as above); the tests hold them bit-exact to the CPU
// restatement of the same path, which is pinned against scikit-image 0.18.3 (tests/golden/skimage_0183_rgb.npz).
namespace ggc {

__global__ void __launch_bounds__(256) k_minmax_u8(size_t n, const uint8_t* __restrict__ img, int32_t* __restrict__ mm) {
    const uint8_t* im = img + (size_t)blockIdx.y * n;
    int lo = 255, hi = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int v = im[i];
        lo = min(lo, v); hi = max(hi, v);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[2 * blockIdx.y], lo); atomicMax(&mm[2 * blockIdx.y + 1], hi); }
}
__global__ void k_minmax_u8_init(int B, int32_t* mm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { mm[2 * i] = 255; mm[2 * i + 1] = 0; }
}

// rgb.astype(float) -> min-max rescale -> rgb2lab, all float64; first pass of the Gaussian (depth axis of length 1) folded in
__global__ void __launch_bounds__(256) k_rgb_lab64(size_t P, const uint8_t* __restrict__ bgr, const int32_t* __restrict__ mm,
                                                   GaussW g, int smooth, double* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const size_t base = ((size_t)blockIdx.y * P + p) * 3;
    const double mn = (double)mm[2 * blockIdx.y], mx = (double)mm[2 * blockIdx.y + 1];
    const double range = mx - mn;
    double lin[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = (double)bgr[base + 2 - c];                              // R, G, B
        v = v - mn; if (mx != mn) v = v / range;
        lin[c] = v > 0.04045 ? det_pow24((v + 0.055) / 1.055) : v / 12.92;
    }
    const double X = (lin[0] * 0.412453 + lin[1] * 0.357580) + lin[2] * 0.180423;
    const double Y = (lin[0] * 0.212671 + lin[1] * 0.715160) + lin[2] * 0.072169;
    const double Z = (lin[0] * 0.019334 + lin[1] * 0.119193) + lin[2] * 0.950227;
    const double t[3] = {X / 0.95047, Y / 1.0, Z / 1.08883};
    double f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = t[c] > 0.008856 ? det_cbrt(t[c]) : 7.787 * t[c] + 16.0 / 116.0;
    double lab[3] = {116.0 * f[1] - 16.0, 500.0 * (f[0] - f[1]), 200.0 * (f[1] - f[2])};
    if (smooth) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {                                      // every tap of the length-1 reflected axis is the pixel itself
            double tmp = lab[c] * g.w[g.r];
            for (int jj = -g.r; jj < 0; ++jj) tmp += (lab[c] + lab[c]) * g.w[jj + g.r];
            lab[c] = tmp;
        }
    }
    out[base + 0] = lab[0]; out[base + 1] = lab[1]; out[base + 2] = lab[2];
}

template <int AXIS>
__global__ void __launch_bounds__(256) k_gauss64(int H, int W, const double* __restrict__ in, GaussW g, double scale,
                                                 int apply_scale, double* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;                        // x * 3 + c
    if (col >= 3 * W) return;
    const int y = blockIdx.y;
    const size_t n = (size_t)H * W * 3;
    const double* im = in + (size_t)blockIdx.z * n;
    const int x = col / 3, c = col - 3 * x;
    const int l = AXIS == 0 ? y : x;
    const int len = AXIS == 0 ? H : W;
    auto at = [&](int idx) -> double {
        const int j = reflect_sym(idx, len);
        return AXIS == 0 ? im[((size_t)j * W + x) * 3 + c] : im[((size_t)y * W + j) * 3 + c];
    };
    double tmp = at(l) * g.w[g.r];
    for (int jj = -g.r; jj < 0; ++jj) tmp += (at(l + jj) + at(l - jj)) * g.w[jj + g.r];
    if (apply_scale) tmp = tmp * scale;
    out[(size_t)blockIdx.z * n + (size_t)y * W * 3 + col] = tmp;
}
__global__ void __launch_bounds__(256) k_scale64(size_t n, const double* __restrict__ in, double scale, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * scale;
}

__device__ __forceinline__ int4 slic_window64(double cy, double cx, const SlicGeom& g) {
    if (cy != cy || cx != cx) return make_int4(0, 0, 0, 0);
    double fy0 = cy - (double)(2 * g.win_y); if (!(fy0 > 0.0)) fy0 = 0.0;
    double fy1 = cy + (double)(2 * g.win_y) + 1.0; if (!(fy1 < (double)g.H)) fy1 = (double)g.H;
    double fx0 = cx - (double)(2 * g.win_x); if (!(fx0 > 0.0)) fx0 = 0.0;
    double fx1 = cx + (double)(2 * g.win_x) + 1.0; if (!(fx1 < (double)g.W)) fx1 = (double)g.W;
    return make_int4((int)fy0, (int)fy1, (int)fx0, (int)fx1);
}
__global__ void k_init_centers64(SlicGeom g, double* __restrict__ centers, int4* __restrict__ bounds) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= g.K) return;
    const size_t o = (size_t)blockIdx.y * g.K + k;
    const double cy = (double)(g.start_y + (k / g.nx) * g.step_y), cx = (double)(g.start_x + (k % g.nx) * g.step_x);
    centers[o * 5 + 0] = cy; centers[o * 5 + 1] = cx;
    centers[o * 5 + 2] = 0.0; centers[o * 5 + 3] = 0.0; centers[o * 5 + 4] = 0.0;
    bounds[o] = slic_window64(cy, cx, g);
}

struct Cand64 { int k, y0, y1, x0, x1; double cy, cx, c0, c1, c2; };

// pixel-centric assignment as k_slic_assign (candidates of a 32x8 tile compacted in ascending k, strict '>'), one pixel per thread
__global__ void __launch_bounds__(256) k_slic_assign64(SlicGeom g, double sw, const double* __restrict__ image,
                                                       const double* __restrict__ centers, const int4* __restrict__ bounds,
                                                       int32_t* __restrict__ labels, int32_t* __restrict__ stale) {
    __shared__ Cand64 cand[256];
    __shared__ int wcount[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 8;
    const int px = tx0 + (tid & 31), py = ty0 + (tid >> 5);
    const bool in = px < g.W && py < g.H;
    const size_t P = (size_t)g.H * g.W;
    const size_t p = (size_t)b * P + (size_t)py * g.W + px;
    double i0 = 0.0, i1 = 0.0, i2 = 0.0;
    if (in) { i0 = image[3 * p]; i1 = image[3 * p + 1]; i2 = image[3 * p + 2]; }
    double best = INFINITY;
    int lab = -1;
    const double* cen = centers + (size_t)b * g.K * 5;
    const int4* bnd = bounds + (size_t)b * g.K;
    for (int base = 0; base < g.K; base += 256) {
        const int k = base + tid;
        bool hit = false;
        int4 bd = make_int4(0, 0, 0, 0);
        if (k < g.K) {
            bd = bnd[k];
            hit = bd.x < ty0 + 8 && bd.y > ty0 && bd.z < tx0 + 32 && bd.w > tx0 && bd.y > bd.x && bd.w > bd.z;
        }
        const unsigned long long bal = __ballot(hit);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) off += wcount[w]; total += wcount[w]; }
        if (hit) {
            Cand64 c;
            c.k = k; c.y0 = bd.x; c.y1 = bd.y; c.x0 = bd.z; c.x1 = bd.w;
            c.cy = cen[k * 5 + 0]; c.cx = cen[k * 5 + 1]; c.c0 = cen[k * 5 + 2]; c.c1 = cen[k * 5 + 3]; c.c2 = cen[k * 5 + 4];
            cand[off + pre] = c;
        }
        __syncthreads();
        for (int i = 0; i < total; ++i) {
            const Cand64& c = cand[i];
            if (!((unsigned)(px - c.x0) < (unsigned)(c.x1 - c.x0) && (unsigned)(py - c.y0) < (unsigned)(c.y1 - c.y0))) continue;
            const double ty = c.cy - (double)py, tx = c.cx - (double)px;
            double dd = (ty * ty + tx * tx) * sw;
            double t = i0 - c.c0, dc = t * t;
            t = i1 - c.c1; dc += t * t;
            t = i2 - c.c2; dc += t * t;
            dd += dc;
            if (best > dd) { best = dd; lab = c.k; }
        }
        __syncthreads();
    }
    if (in) { if (lab >= 0) labels[p] = lab; else atomicOr(&stale[b], 1); }
}

// centre update as k_slic_update (8-lane group per cluster, ordered colour fold), sums in double
__global__ void __launch_bounds__(256) k_slic_update64(SlicGeom g, const double* __restrict__ image, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ stale, double* __restrict__ centers,
                                                       int4* __restrict__ bounds) {
    const int lane = threadIdx.x & 63, sub = lane / UPD_LANES, sl = lane % UPD_LANES;
    const int k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * UPD_GROUPS + sub;
    const bool live = k < g.K;
    const int b = blockIdx.y;
    const size_t o = (size_t)b * g.K + (live ? k : 0);
    int4 bd = live ? bounds[o] : make_int4(0, 0, 0, 0);
    if (live && stale[b]) bd = make_int4(0, g.H, 0, g.W);
    const size_t P = (size_t)g.H * g.W;
    const int32_t* lb = labels + (size_t)b * P;
    const double* im = image + (size_t)b * P * 3;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    long long cnt = 0, isy = 0, isx = 0;                    // integer coordinate sums: exact in double far beyond any image
    const int rows = bd.y - bd.x, cols = bd.w - bd.z;
    int rows_max = rows, cols_max = cols;
#pragma unroll
    for (int off = UPD_LANES; off < 64; off <<= 1) {
        rows_max = max(rows_max, __shfl_xor(rows_max, off, 64));
        cols_max = max(cols_max, __shfl_xor(cols_max, off, 64));
    }
    for (int ry = 0; ry < rows_max; ++ry) {
        const int y = bd.x + ry;
        for (int c0 = 0; c0 < cols_max; c0 += UPD_LANES) {
            const int x = bd.z + c0 + sl;
            const bool hit = live && ry < rows && x < bd.w && lb[(size_t)y * g.W + x] == k;
            const unsigned long long bal = __ballot(hit);
            if (!bal) continue;
            unsigned int m = (unsigned int)(bal >> (sub * UPD_LANES)) & ((1u << UPD_LANES) - 1u);
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            if (hit) { const double* px = im + ((size_t)y * g.W + x) * 3; v0 = px[0]; v1 = px[1]; v2 = px[2]; isx += x; }
            const int n = __popc(m);
            cnt += n; isy += (long long)n * y;
            while (__any(m != 0u)) {
                const int src = sub * UPD_LANES + (m ? __ffs((int)m) - 1 : 0);
                const double t0 = __shfl(v0, src, 64), t1 = __shfl(v1, src, 64), t2 = __shfl(v2, src, 64);
                if (m) { s0 += t0; s1 += t1; s2 += t2; }
                m &= m - 1u;
            }
        }
    }
#pragma unroll
    for (int off = 1; off < UPD_LANES; off <<= 1) isx += __shfl_xor(isx, off, 64);
    if (!live || sl != 0) return;
    const double n = (double)cnt;
    const double cy = (double)isy / n, cx = (double)isx / n;
    centers[o * 5 + 0] = cy; centers[o * 5 + 1] = cx;
    centers[o * 5 + 2] = s0 / n; centers[o * 5 + 3] = s1 / n; centers[o * 5 + 4] = s2 / n;
    bounds[o] = slic_window64(cy, cx, g);
}

} // namespace ggc

extern "C" int ggc_slic_rgb(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const uint8_t* bgr, int n_segments,
                            double compactness, double sigma, int32_t* segments, int32_t* n_nodes) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535 && H <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, bgr && segments && n_nodes, GGC_E_INVALID_ARG, "null pointer");
    GGC_REQUIRE(ctx, n_segments >= 1 && compactness > 0.0 && sigma >= 0.0, GGC_E_INVALID_ARG,
                "bad SLIC parameters n_segments=%d compactness=%g sigma=%g", n_segments, compactness, sigma);
    GGC_REQUIRE(ctx, (size_t)H * W < (1u << 30), GGC_E_SHAPE, "image too large");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t P = (size_t)H * W;
    const Grid seed = regular_grid(H, W, n_segments);
    GGC_REQUIRE(ctx, seed.K >= 1, GGC_E_SHAPE, "no SLIC seeds for %dx%d with n_segments=%d", H, W, n_segments);
    const Grid win = regular_grid(H, W, seed.K);
    SlicGeom g{};
    g.H = H; g.W = W; g.K = seed.K;
    g.ny = seed.ny; g.nx = seed.nx; g.start_y = seed.start_y; g.start_x = seed.start_x;
    g.step_y = seed.step_y; g.step_x = seed.step_x;
    g.win_y = win.step_y; g.win_x = win.step_x;
    const double step = (double)std::max(std::max(seed.step_y, seed.step_x), 1);
    const double sw = 1.0 / (step * step);

    double* img_a = scratch_t<double>(ctx, S_SLIC_IMG, (size_t)B * P * 3);
    double* img_b = scratch_t<double>(ctx, S_SLIC_TMP, (size_t)B * P * 3);
    int32_t* mm = scratch_t<int32_t>(ctx, S_SLIC_MINMAX, (size_t)B * 2);
    double* centers = scratch_t<double>(ctx, S_SLIC_CENTERS, (size_t)B * g.K * 5);
    int4* bounds = scratch_t<int4>(ctx, S_SLIC_AUX, (size_t)B * g.K);
    int32_t* raw = scratch_t<int32_t>(ctx, S_SLIC_LABELS, (size_t)B * P);
    int32_t* stale = scratch_t<int32_t>(ctx, S_SLIC_AUX2, (size_t)B * 10);
    if (!img_a || !img_b || !mm || !centers || !bounds || !raw || !stale) return GGC_E_OOM;

    GaussW gw{};
    const int smooth = sigma > 0.0 ? 1 : 0;
    if (smooth) {
        gw.r = (int)(4.0 * sigma + 0.5);
        GGC_REQUIRE(ctx, gw.r <= MAX_RADIUS, GGC_E_UNSUPPORTED, "sigma=%g needs radius %d > %d", sigma, gw.r, MAX_RADIUS);
        ggc_gaussian_taps(sigma, gw.r, gw.w);
        const double sum = np_pairwise_sum(gw.w, 2 * gw.r + 1);
        for (int i = 0; i < 2 * gw.r + 1; ++i) gw.w[i] = gw.w[i] / sum;
    }
    hipLaunchKernelGGL(k_minmax_u8_init, dim3(cdiv(B, 256)), dim3(256), 0, st, B, mm);
    hipLaunchKernelGGL(k_minmax_u8, dim3(std::min(cdiv(P * 3, 256 * 8), 48), B), dim3(256), 0, st, P * 3, bgr, mm);
    hipLaunchKernelGGL(k_rgb_lab64, dim3(cdiv(P, 256), B), dim3(256), 0, st, P, bgr, mm, gw, smooth, img_a);
    GGC_LAUNCH_CHECK(ctx);
    const double ratio = 1.0 / compactness;
    const double* km_img;
    if (smooth) {
        const dim3 grid(cdiv((size_t)W * 3, 256), H, B);
        hipLaunchKernelGGL((k_gauss64<0>), grid, dim3(256), 0, st, H, W, img_a, gw, 1.0, 0, img_b);
        hipLaunchKernelGGL((k_gauss64<1>), grid, dim3(256), 0, st, H, W, img_b, gw, ratio, 1, img_a);
        km_img = img_a;
    } else {
        hipLaunchKernelGGL(k_scale64, dim3(cdiv((size_t)B * P * 3, 256)), dim3(256), 0, st, (size_t)B * P * 3, img_a, ratio, img_b);
        km_img = img_b;
    }
    GGC_LAUNCH_CHECK(ctx);
    GGC_HIP(ctx, hipMemsetAsync(raw, 0, sizeof(int32_t) * (size_t)B * P, st));
    GGC_HIP(ctx, hipMemsetAsync(stale, 0, sizeof(int32_t) * (size_t)B * 10, st));
    hipLaunchKernelGGL(k_init_centers64, dim3(cdiv(g.K, 256), B), dim3(256), 0, st, g, centers, bounds);
    for (int it = 0; it < 10; ++it) {
        hipLaunchKernelGGL(k_slic_assign64, dim3(cdiv(W, 32), cdiv(H, 8), B), dim3(256), 0, st, g, sw, km_img, centers, bounds, raw,
                           stale + (size_t)it * B);
        hipLaunchKernelGGL(k_slic_update64, dim3(cdiv(g.K, 4 * UPD_GROUPS), B), dim3(256), 0, st, g, km_img, raw, stale + (size_t)it * B,
                           centers, bounds);
        GGC_LAUNCH_CHECK(ctx);
    }
    const double seg_size = (double)P / (double)g.K;
    return enforce_connectivity(ctx, st, B, H, W, raw, (int)(0.5 * seg_size), (int)(3.0 * seg_size), segments, n_nodes);
}

// Step 7 alone (skimage's _enforce_label_connectivity_cython) — exposed so the parity
// tests can drive the carve / merge paths with hand-made label maps.
extern "C" int ggc_slic_enforce_connectivity(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W,
                                             const int32_t* raw_labels, int min_size, int max_size,
                                             int32_t* segments, int32_t* n_nodes) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, raw_labels && segments && n_nodes && min_size >= 0 && max_size >= 0, GGC_E_INVALID_ARG, "bad arguments");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    return enforce_connectivity(ctx, reinterpret_cast<hipStream_t>(stream), B, H, W, raw_labels, min_size, max_size,
                                segments, n_nodes);
}
