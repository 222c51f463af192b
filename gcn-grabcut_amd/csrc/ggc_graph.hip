// ggc_graph.hip — G2-G8: region statistics, node features, region-adjacency +
// non-local colour edges, automatic prior (reference graph_builder.py:190-454).
//
// Batched over B images with data-dependent node counts.  Design points:
//  * region sums follow np.bincount(weights=...) exactly — float64 running sums
//    in raster order, cast to float32 — by giving every region one lane that
//    walks the region's bounding box in raster order (SLIC regions are compact,
//    so the box is a few thousand pixels).  Deterministic, no float atomics.
//  * np.unique(lo*N+hi, return_counts=True) becomes a dense per-image N x N
//    int32 matrix filled with integer atomics (exact, order independent) and
//    read back row-major with wave ballot/prefix compaction, which yields the
//    pairs already sorted by code.  288 GB of HBM makes the dense form cheap:
//    256 images x 700^2 x 4 B = 0.5 GB.
//  * the non-local k-NN (argpartition) is one wave per node over an LDS row of
//    distances: k rounds of wave arg-min with ties to the lowest index.
//  * ggc_graph_count synchronises twice (node counts, pair counts) so that the
//    packed outputs can be sized exactly; ggc_graph_fill only copies.
#include "ggc_internal.h"
#include <atomic>
#include <algorithm>
#include <cmath>
#include <vector>

namespace ggc {

constexpr int NL_BIT = 1 << 30;

__device__ __forceinline__ uint32_t f2ord_g(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f_g(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

struct GDims { int B, H, W, Nmax, conn, k_nl; };

// Per-region record produced by k_stats (all float32, as the reference stores them).
struct RegionStats {
    float cnt, safe;
    float mlab[3], slab[3], mhsv[3];
    float cy, cx;         // _region_statistics centroids (f32 y/H accumulated in f64)
    float pcy, pcx;       // compute_auto_prior centroids (f64 y/H)
    float bpx, mgrad, mgn, area, border;
};

// ---- K1: bounding boxes, frame counts, per-image max gradient
constexpr int BBOX_ROWS = 32;
__global__ void __launch_bounds__(256) k_bbox(GDims d, const int32_t* __restrict__ seg, const float* __restrict__ grad,
                                              int4* __restrict__ bbox, int32_t* __restrict__ border,
                                              uint32_t* __restrict__ gmax) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b = blockIdx.z;
    float gv = -INFINITY;
    const int lane = threadIdx.x & 63;
    // a block walks BBOX_ROWS rows: 4 rows per block made the launch itself (134 k tiny blocks) the cost
    for (int y = blockIdx.y * BBOX_ROWS + (threadIdx.x >> 6); y < min((int)(blockIdx.y + 1) * BBOX_ROWS, d.H); y += 4) {
    const bool valid = x < d.W;
    int s = -1;
    size_t p = 0;
    if (valid) {
        p = (size_t)b * d.H * d.W + (size_t)y * d.W + x;
        s = seg[p];
        const int mult = (y == 0) + (y == d.H - 1) + (x == 0) + (x == d.W - 1);
        if (mult) atomicAdd(&border[(size_t)b * d.Nmax + s], mult);
        gv = fmaxf(gv, grad[p]);
    }
    // A region's pixels inside this wave's 64-pixel row segment form runs; a run's x extremes are its two ends,
    // so its first lane issues the four min/max atomics for the whole run (4 per run instead of 4 per outline pixel).
    const int s_left = __shfl_up(s, 1, 64);
    const unsigned long long vmask = __ballot(valid);
    const unsigned long long starts = __ballot(valid && (lane == 0 || s_left != s));
    if (valid && ((starts >> lane) & 1ull)) {
        const unsigned long long stop = (starts | ~vmask) >> lane >> 1;
        const int len = stop ? __ffsll((long long)stop) : 64 - lane;
        int4* bb = bbox + (size_t)b * d.Nmax + s;
        // ... and a run whose neighbour row holds a same-region run reaching at least as far cannot set that extreme:
        // only "corner" runs reach the L2 atomic units (their throughput is what bounds this kernel)
        const int xe = x + len - 1;
        const bool up = y > 0, dn = y + 1 < d.H;
        const bool up_s = up && seg[p - d.W] == s, dn_s = dn && seg[p + d.W] == s;
        const bool up_e = up && seg[p - d.W + len - 1] == s, dn_e = dn && seg[p + d.W + len - 1] == s;
        if (!up_s && !up_e) atomicMin(&bb->x, y);
        if (!dn_s && !dn_e) atomicMax(&bb->y, y + 1);
        const bool left_cov = x > 0 && ((up_s && seg[p - d.W - 1] == s) || (dn_s && seg[p + d.W - 1] == s));
        const bool right_cov = xe + 1 < d.W && ((up_e && seg[p - d.W + len] == s) || (dn_e && seg[p + d.W + len] == s));
        if (!left_cov) atomicMin(&bb->z, x);
        if (!right_cov) atomicMax(&bb->w, xe + 1);
    }
    }
    for (int o = 32; o > 0; o >>= 1) gv = fmaxf(gv, __shfl_xor(gv, o, 64));
    if ((threadIdx.x & 63) == 0 && gv > -INFINITY) atomicMax(&gmax[b], f2ord_g(gv));
}

__global__ void k_bbox_init(size_t n, int4* bbox) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bbox[i] = make_int4(INT32_MAX, 0, INT32_MAX, 0);
}

// ---- K2: region statistics, one lane per region, raster order, f64 sums
// The sums must be float64 running sums in raster order (= np.bincount(weights=...)), so a lane owns a region and walks
// its bounding box.  Read straight from global memory that walk moved 15x the bytes it needed (PMC: 15.8 GB per launch
// against 1 GB of inputs — 64 lanes, 64 different cache lines per load).  Instead the wave stages one image ROW of the
// union of its 64 regions' boxes in LDS with coalesced loads (labels of the rows above / below too, for the boundary
// test), and every lane scans its own x-range of that row from LDS.  Same visits in the same order, every byte loaded
// about twice (overlap of neighbouring waves' unions).
struct StatsAcc {
    double cnt = 0, sl[3] = {0, 0, 0}, sl2[3] = {0, 0, 0}, sh[3] = {0, 0, 0};
    double sy = 0, sx = 0, syd = 0, sxd = 0, sb = 0, sg1 = 0, sgn = 0;
};

__global__ void __launch_bounds__(64) k_stats(GDims d, const int32_t* __restrict__ seg,
                                              const int32_t* __restrict__ n_nodes,
                                              const float* __restrict__ lab, const float* __restrict__ hsv,
                                              const float* __restrict__ grad, const int4* __restrict__ bbox,
                                              const int32_t* __restrict__ border,
                                              const uint32_t* __restrict__ gmax, RegionStats* __restrict__ out) {
    extern __shared__ float s_row[];          // [3][Wu] labels (rows y-1, y, y+1) | [Wu][3] lab | [Wu][3] hsv | [Wu] grad | [Wu] grad/max
                                              // | [Wu] (float)x/W | [Wu] (double)x/W: the quotients are per pixel / per column, not per visit
    const int b = blockIdx.y, lane = threadIdx.x;
    const int N = n_nodes[b];
    const int r0 = blockIdx.x * 64;
    if (r0 >= N) return;
    const int r = r0 + lane;
    const bool live = r < N;
    const int H = d.H, W = d.W;
    const size_t P = (size_t)H * W;
    const int32_t* sg = seg + (size_t)b * P;
    const float* lb = lab + (size_t)b * P * 3;
    const float* hv = hsv + (size_t)b * P * 3;
    const float* gr = grad + (size_t)b * P;
    const int4 bb = live ? bbox[(size_t)b * d.Nmax + r] : make_int4(INT32_MAX, 0, INT32_MAX, 0);
    const float gden = (float)((double)ord2f_g(gmax[b]) + 1e-6);
    // union of the wave's boxes
    int uy0 = bb.y > bb.x ? bb.x : INT32_MAX, uy1 = bb.y > bb.x ? bb.y : 0;
    int ux0 = bb.w > bb.z ? bb.z : INT32_MAX, ux1 = bb.w > bb.z ? bb.w : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uy0 = min(uy0, __shfl_xor(uy0, o, 64)); uy1 = max(uy1, __shfl_xor(uy1, o, 64));
        ux0 = min(ux0, __shfl_xor(ux0, o, 64)); ux1 = max(ux1, __shfl_xor(ux1, o, 64));
    }
    const int Wu = max(ux1 - ux0, 0);
    int32_t* s_seg = reinterpret_cast<int32_t*>(s_row);            // 3 rows, slot = (y + 1) % 3 rotates
    float* s_lab = s_row + 3 * (size_t)Wu;
    float* s_hsv = s_lab + 3 * (size_t)Wu;
    float* s_grd = s_hsv + 3 * (size_t)Wu;
    float* s_gn = s_grd + Wu;
    float* s_xf = s_gn + Wu;
    double* s_xd = reinterpret_cast<double*>(s_xf + Wu + (Wu & 1));          // 8-byte aligned: 12 Wu + (Wu & 1) floats precede it
    for (int i = lane; i < Wu; i += 64) { s_xf[i] = (float)(ux0 + i) / (float)W; s_xd[i] = (double)(ux0 + i) / (double)W; }
    // Rows are staged by LDS-DMA (global_load_lds, 4 B per lane, 256 B per instruction): a single wave cannot hide
    // the latency of load -> store loops, but it can have a whole row of DMA pieces in flight and wait once.
    auto dma_row = [&](const void* src, void* dst, int n_words) {   // n_words 4-byte words, contiguous on both sides
        for (int i = 0; i < n_words; i += 64)
            if (i + lane < n_words)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const uint32_t*)src + i + lane),
                                                 (__attribute__((address_space(3))) void*)((uint32_t*)dst + i), 4, 0, 0);
    };
    auto load_seg_row = [&](int y) {                                // labels of image row y (or -1 outside) into its slot
        int32_t* dst = s_seg + (size_t)((y + 3) % 3) * Wu;
        if (y >= 0 && y < H) dma_row(sg + (size_t)y * W + ux0, dst, Wu);
        else for (int i = lane; i < Wu; i += 64) dst[i] = -1;
    };
    StatsAcc a;
    if (uy1 > uy0 && Wu > 0) {
        load_seg_row(uy0 - 1);
        load_seg_row(uy0);
        for (int y = uy0; y < uy1; ++y) {
            load_seg_row(y + 1);
            const size_t row = (size_t)y * W + ux0;
            dma_row(lb + 3 * row, s_lab, 3 * Wu);
            dma_row(hv + 3 * row, s_hsv, 3 * Wu);
            dma_row(gr + row, s_grd, Wu);
            const double yf = (double)((float)y / (float)H), yd = (double)y / (double)H;
            __builtin_amdgcn_s_waitcnt(0);                          // every DMA piece of this row has landed
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < Wu; i += 64) s_gn[i] = s_grd[i] / gden;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (live && y >= bb.x && y < bb.y) {
                const int32_t* cur = s_seg + (size_t)((y + 3) % 3) * Wu;
                const int32_t* up = s_seg + (size_t)((y + 2) % 3) * Wu;
                const int32_t* dn = s_seg + (size_t)((y + 4) % 3) * Wu;
                for (int x = bb.z; x < bb.w; ++x) {
                    const int i = x - ux0;
                    if (cur[i] != r) continue;
                    a.cnt += 1.0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = s_lab[3 * i + c];
                        a.sl[c] += (double)v;
                        a.sl2[c] += (double)(v * v);
                        a.sh[c] += (double)s_hsv[3 * i + c];
                    }
                    a.sy += yf;
                    a.sx += (double)s_xf[i];
                    a.syd += yd;
                    a.sxd += s_xd[i];
                    // find_boundaries(mode="inner"): 4-neighbourhood max != min, and label != 0
                    int mx = r, mn = r;
                    if (y > 0) { const int u = up[i]; mx = max(mx, u); mn = min(mn, u); }
                    if (y < H - 1) { const int u = dn[i]; mx = max(mx, u); mn = min(mn, u); }
                    // the left / right neighbours can lie just outside the staged columns: read those from memory
                    if (x > 0) { const int u = i > 0 ? cur[i - 1] : sg[(size_t)y * W + x - 1]; mx = max(mx, u); mn = min(mn, u); }
                    if (x < W - 1) { const int u = i + 1 < Wu ? cur[i + 1] : sg[(size_t)y * W + x + 1]; mx = max(mx, u); mn = min(mn, u); }
                    if (mx != mn && r != 0) a.sb += 1.0;
                    a.sg1 += (double)s_grd[i];
                    a.sgn += (double)s_gn[i];
                }
            }
            __builtin_amdgcn_wave_barrier();       // everyone is done with this row before its buffers are overwritten
        }
    }
    if (!live) return;
    RegionStats s;
    s.cnt = (float)a.cnt;
    s.safe = s.cnt > 1.0f ? s.cnt : 1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float m = (float)a.sl[c] / s.safe;
        const float sq = (float)a.sl2[c] / s.safe;
        float v = sq - m * m;
        if (!(v > 0.0f)) v = (v != v) ? v : 0.0f;
        s.mlab[c] = m; s.slab[c] = sqrtf(v);
        s.mhsv[c] = (float)a.sh[c] / s.safe;
    }
    s.cy = (float)a.sy / s.safe; s.cx = (float)a.sx / s.safe;
    s.pcy = (float)(a.syd / (double)s.safe); s.pcx = (float)(a.sxd / (double)s.safe);
    s.bpx = (float)a.sb;
    s.mgrad = (float)a.sg1 / s.safe;
    s.mgn = (float)a.sgn / s.safe;
    s.area = s.cnt / (float)((double)H * (double)W);
    s.border = (float)border[(size_t)b * d.Nmax + r];
    out[(size_t)b * d.Nmax + r] = s;
}

// ---- K3: adjacency counts into the dense matrix (integer atomics: exact)
__global__ void __launch_bounds__(256) k_adj(GDims d, const int32_t* __restrict__ seg, int32_t* __restrict__ dense) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= d.W || y >= d.H) return;
    const int b = blockIdx.z;
    const int32_t* sg = seg + (size_t)b * d.H * d.W;
    int32_t* m = dense + (size_t)b * d.Nmax * d.Nmax;
    const int a = sg[(size_t)y * d.W + x];
    auto link = [&](int u, int v) {
        if (u != v) atomicAdd(&m[(size_t)min(u, v) * d.Nmax + max(u, v)], 1);
    };
    if (x + 1 < d.W) link(a, sg[(size_t)y * d.W + x + 1]);
    if (y + 1 < d.H) link(a, sg[(size_t)(y + 1) * d.W + x]);
    if (d.conn == 8 && x + 1 < d.W && y + 1 < d.H) {
        link(a, sg[(size_t)(y + 1) * d.W + x + 1]);
        link(sg[(size_t)y * d.W + x + 1], sg[(size_t)(y + 1) * d.W + x]);
    }
}

// ---- K6: non-local k nearest neighbours in mean-Lab (one wave per node)
__global__ void __launch_bounds__(256) k_knn(GDims d, const int32_t* __restrict__ n_nodes,
                                             const RegionStats* __restrict__ st, int32_t* __restrict__ dense) {
    extern __shared__ float drow[];               // 4 waves x Nmax distances
    const int b = blockIdx.y;
    const int N = n_nodes[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (!(d.k_nl > 0 && N > d.k_nl + 1) || i >= N) return;   // graph_builder.py:291
    float* dr = drow + (size_t)wave * d.Nmax;
    const RegionStats* s = st + (size_t)b * d.Nmax;
    int32_t* m = dense + (size_t)b * d.Nmax * d.Nmax;
    const float l0 = s[i].mlab[0], l1 = s[i].mlab[1], l2 = s[i].mlab[2];
    for (int j = lane; j < N; j += 64) {
        const float dx = l0 - s[j].mlab[0], dy = l1 - s[j].mlab[1], dz = l2 - s[j].mlab[2];
        float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
        const int lo = min(i, j), hi = max(i, j);
        if (j == i || (m[(size_t)lo * d.Nmax + hi] & ~NL_BIT) != 0) dist = INFINITY;
        dr[j] = dist;
    }
    // (same wave wrote and reads dr: no barrier needed beyond the implicit wave order)
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < d.k_nl; ++t) {
        float best = NAN; int bj = INT32_MAX;       // arg-min over non-taken entries (NaN = taken)
        for (int j = lane; j < N; j += 64) {
            const float v = dr[j];
            if (v == v && (bj == INT32_MAX || v < best)) { best = v; bj = j; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            const bool take = (oj != INT32_MAX) && (bj == INT32_MAX || ov < best || (ov == best && oj < bj));
            if (take) { best = ov; bj = oj; }
        }
        if (bj == INT32_MAX) break;
        if (lane == 0) {
            dr[bj] = NAN;
            if (isfinite(best)) atomicOr(&m[(size_t)min(i, bj) * d.Nmax + max(i, bj)], NL_BIT);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- K4/K7: per-row counts of adjacency / non-local entries (wave per row)
__global__ void __launch_bounds__(256) k_rowcount(GDims d, const int32_t* __restrict__ n_nodes,
                                                  const int32_t* __restrict__ dense,
                                                  int32_t* __restrict__ cnt_adj, int32_t* __restrict__ cnt_nl) {
    const int b = blockIdx.y;
    const int N = n_nodes[b];
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int32_t* row = dense + ((size_t)b * d.Nmax + i) * d.Nmax;
    int ca = 0, cn = 0;
    for (int j = i + 1 + lane; j < N; j += 64) {
        const int v = row[j];
        ca += (v & ~NL_BIT) != 0;
        cn += (v & NL_BIT) != 0;
    }
    for (int o = 32; o > 0; o >>= 1) { ca += __shfl_xor(ca, o, 64); cn += __shfl_xor(cn, o, 64); }
    if (lane == 0) { cnt_adj[(size_t)b * d.Nmax + i] = ca; cnt_nl[(size_t)b * d.Nmax + i] = cn; }
}

// per-image exclusive scans of the two row-count arrays; totals[b] = {n_adj, n_nl, max shared}
__global__ void __launch_bounds__(256) k_rowscan(GDims d, const int32_t* __restrict__ n_nodes,
                                                 int32_t* __restrict__ cnt_adj, int32_t* __restrict__ cnt_nl,
                                                 int32_t* __restrict__ totals) {
    __shared__ int pa[256], pn[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = n_nodes[b];
    int32_t* ca = cnt_adj + (size_t)b * d.Nmax;
    int32_t* cn = cnt_nl + (size_t)b * d.Nmax;
    const int chunk = (N + 255) / 256;
    const int beg = min(tid * chunk, N), end = min(beg + chunk, N);
    int sa = 0, sn = 0;
    for (int i = beg; i < end; ++i) { sa += ca[i]; sn += cn[i]; }
    pa[tid] = sa; pn[tid] = sn;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int va = tid >= off ? pa[tid - off] : 0, vn = tid >= off ? pn[tid - off] : 0;
        __syncthreads();
        pa[tid] += va; pn[tid] += vn;
        __syncthreads();
    }
    int ra = tid ? pa[tid - 1] : 0, rn = tid ? pn[tid - 1] : 0;
    for (int i = beg; i < end; ++i) { const int a = ca[i], n = cn[i]; ca[i] = ra; cn[i] = rn; ra += a; rn += n; }
    if (tid == 255) { totals[3 * b] = pa[255]; totals[3 * b + 1] = pn[255]; }
}

// ---- K5: ordered extraction of the pair lists (wave per row), raw pair features
struct PairOut {
    int32_t* lo; int32_t* hi; float* de; float* dxy; float* shared; float* gc;
};

__global__ void __launch_bounds__(256) k_extract(GDims d, const int32_t* __restrict__ n_nodes,
                                                 const int32_t* __restrict__ dense,
                                                 const RegionStats* __restrict__ st,
                                                 const int32_t* __restrict__ off_adj, const int32_t* __restrict__ off_nl,
                                                 const int64_t* __restrict__ pair_ptr /*[B+1] adj+nl pairs*/,
                                                 const int32_t* __restrict__ totals, PairOut out,
                                                 uint32_t* __restrict__ maxima /*[B,5]: de_adj,dxy_adj,de_nl,dxy_nl,shared*/) {
    const int b = blockIdx.y;
    const int N = n_nodes[b];
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int32_t* row = dense + ((size_t)b * d.Nmax + i) * d.Nmax;
    const RegionStats* s = st + (size_t)b * d.Nmax;
    const int64_t base_adj = pair_ptr[b] + off_adj[(size_t)b * d.Nmax + i];
    const int64_t base_nl = pair_ptr[b] + totals[3 * b] + off_nl[(size_t)b * d.Nmax + i];
    int run_a = 0, run_n = 0;
    float mde_a = 0.f, mdx_a = 0.f, mde_n = 0.f, mdx_n = 0.f;
    int msh = 0;
    const RegionStats si = s[i];
    for (int j0 = i + 1; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const int v = j < N ? row[j] : 0;
        const bool isa = (v & ~NL_BIT) != 0, isn = (v & NL_BIT) != 0;
        const unsigned long long ba = __ballot(isa), bn = __ballot(isn);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (isa || isn) {
            const RegionStats sj = s[j];
            const float dx = si.mlab[0] - sj.mlab[0], dy = si.mlab[1] - sj.mlab[1], dz = si.mlab[2] - sj.mlab[2];
            const float de = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float cy = si.cy - sj.cy, cx = si.cx - sj.cx;
            const float dxy = sqrtf(cy * cy + cx * cx);
            const float gc = fabsf(si.mgn - sj.mgn);
            if (isa) {
                const int64_t q = base_adj + run_a + __popcll(ba & lt);
                out.lo[q] = i; out.hi[q] = j; out.de[q] = de; out.dxy[q] = dxy; out.gc[q] = gc;
                out.shared[q] = (float)(v & ~NL_BIT);
                mde_a = fmaxf(mde_a, de); mdx_a = fmaxf(mdx_a, dxy); msh = max(msh, v & ~NL_BIT);
            }
            if (isn) {
                const int64_t q = base_nl + run_n + __popcll(bn & lt);
                out.lo[q] = i; out.hi[q] = j; out.de[q] = de; out.dxy[q] = dxy; out.gc[q] = gc;
                out.shared[q] = 0.0f;
                mde_n = fmaxf(mde_n, de); mdx_n = fmaxf(mdx_n, dxy);
            }
        }
        run_a += __popcll(ba); run_n += __popcll(bn);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mde_a = fmaxf(mde_a, __shfl_xor(mde_a, o, 64)); mdx_a = fmaxf(mdx_a, __shfl_xor(mdx_a, o, 64));
        mde_n = fmaxf(mde_n, __shfl_xor(mde_n, o, 64)); mdx_n = fmaxf(mdx_n, __shfl_xor(mdx_n, o, 64));
        msh = max(msh, __shfl_xor(msh, o, 64));
    }
    if (lane == 0) {
        uint32_t* mx = maxima + (size_t)b * 5;
        if (run_a) { atomicMax(&mx[0], f2ord_g(mde_a)); atomicMax(&mx[1], f2ord_g(mdx_a)); atomicMax(&mx[4], (uint32_t)msh); }
        if (run_n) { atomicMax(&mx[2], f2ord_g(mde_n)); atomicMax(&mx[3], f2ord_g(mdx_n)); }
    }
}

// ---- K9: final edge attributes in place (de, dxy, shared normalised per image and per pair kind)
__global__ void __launch_bounds__(256) k_pair_final(int B, const int64_t* __restrict__ pair_ptr,
                                                    const int32_t* __restrict__ totals,
                                                    const uint32_t* __restrict__ maxima, PairOut io,
                                                    float* __restrict__ flag) {
    const int b = blockIdx.y;
    const int64_t beg = pair_ptr[b], end = pair_ptr[b + 1];
    const int64_t q = beg + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= end) return;
    const bool nl = (q - beg) >= totals[3 * b];
    const uint32_t* mx = maxima + (size_t)b * 5;
    const float de_den = (float)((double)ord2f_g(mx[nl ? 2 : 0]) + 1e-6);
    const float dx_den = (float)((double)ord2f_g(mx[nl ? 3 : 1]) + 1e-6);
    io.de[q] = io.de[q] / de_den;
    io.dxy[q] = io.dxy[q] / dx_den;
    if (!nl) io.shared[q] = io.shared[q] / (float)((double)mx[4] + 1e-6);
    flag[q] = nl ? 1.0f : 0.0f;
}

// ---- K10: node features incl. per-image min-max of the colour columns (block per image)
__device__ __forceinline__ float fix_nan(float v) {
    if (v != v) return 0.0f;
    if (isinf(v)) return v > 0 ? 1.0f : 0.0f;
    return v;
}

__global__ void __launch_bounds__(256) k_node_feats(GDims d, const int32_t* __restrict__ n_nodes,
                                                    const RegionStats* __restrict__ st,
                                                    float* __restrict__ feat /*[B,Nmax,16]*/) {
    __shared__ float rmin[6][256], rmax[6][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = n_nodes[b];
    const RegionStats* s = st + (size_t)b * d.Nmax;
    float* f = feat + (size_t)b * d.Nmax * 16;
    float lo[6], hi[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { lo[c] = INFINITY; hi[c] = -INFINITY; }
    for (int i = tid; i < N; i += 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], s[i].mlab[c]); hi[c] = fmaxf(hi[c], s[i].mlab[c]);
            lo[3 + c] = fminf(lo[3 + c], s[i].slab[c]); hi[3 + c] = fmaxf(hi[3 + c], s[i].slab[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) { rmin[c][tid] = lo[c]; rmax[c][tid] = hi[c]; }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                rmin[c][tid] = fminf(rmin[c][tid], rmin[c][tid + o]);
                rmax[c][tid] = fmaxf(rmax[c][tid], rmax[c][tid + o]);
            }
        __syncthreads();
    }
    const float four_pi = (float)(4 * 3.141592653589793);
    for (int i = tid; i < N; i += 256) {
        const RegionStats r = s[i];
        float* o = f + (size_t)i * 16;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = fix_nan((r.mlab[c] - rmin[c][0]) / ((rmax[c][0] - rmin[c][0]) + (float)1e-6));
            o[3 + c] = fix_nan((r.slab[c] - rmin[3 + c][0]) / ((rmax[3 + c][0] - rmin[3 + c][0]) + (float)1e-6));
            o[6 + c] = fix_nan(r.mhsv[c]);
        }
        o[9] = fix_nan(r.cy); o[10] = fix_nan(r.cx); o[11] = fix_nan(r.area);
        const float per = r.bpx > 1.0f ? r.bpx : 1.0f;
        float comp = (four_pi * r.cnt) / (per * per);
        comp = comp < 0.0f ? 0.0f : (comp > 1.0f ? 1.0f : comp);
        o[12] = fix_nan(comp);
        o[13] = fix_nan(r.mgrad / (float)255.0);
        o[14] = fix_nan(r.bpx / r.safe);
        const float a = r.cy - 0.5f, c2 = r.cx - 0.5f;
        o[15] = fix_nan(sqrtf(a * a + c2 * c2) / (float)0.707);
    }
}

// ---- K11: prior cue 1, raw contrast (wave per node)
__global__ void __launch_bounds__(256) k_contrast(GDims d, const int32_t* __restrict__ n_nodes,
                                                  const RegionStats* __restrict__ st, float two_cs2, float* __restrict__ contrast) {
    const int b = blockIdx.y;
    const int N = n_nodes[b];
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const RegionStats* s = st + (size_t)b * d.Nmax;
    const float csum = fmaxf((float)(d.H * d.W), 1.0f);   // counts.sum() == H*W exactly in float32 (< 2^24)
    const RegionStats si = s[i];
    float acc = 0.0f;
    for (int j = lane; j < N; j += 64) {
        const float dx = si.mlab[0] - s[j].mlab[0], dy = si.mlab[1] - s[j].mlab[1], dz = si.mlab[2] - s[j].mlab[2];
        const float cd = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float a = si.pcy - s[j].pcy, c = si.pcx - s[j].pcx;
        const float sd = sqrtf(a * a + c * c);
        acc += (cd * ggc_expf(-(sd * sd) / two_cs2)) * (s[j].cnt / csum);
    }
    acc = wave_sum(acc);
    if (lane == 0) contrast[(size_t)b * d.Nmax + i] = acc;
}

// block-wide min/max helper over values v(i), i < N
template <typename F>
__device__ void block_minmax(int N, F v, float& mn, float& mx, float* sa, float* sb) {
    const int tid = threadIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < N; i += 256) { const float x = v(i); lo = fminf(lo, x); hi = fmaxf(hi, x); }
    sa[tid] = lo; sb[tid] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sa[tid] = fminf(sa[tid], sa[tid + o]); sb[tid] = fmaxf(sb[tid], sb[tid + o]); }
        __syncthreads();
    }
    mn = sa[0]; mx = sb[0];
    __syncthreads();
}

__device__ __forceinline__ float unit_apply(float v, float mn, float mx) {  // _unit_norm (:447-454)
    if ((double)mx - (double)mn < 1e-8) return 0.0f;
    return (v - mn) / (float)((double)mx - (double)mn);
}

// ---- K12: the rest of compute_auto_prior (block per image)
__global__ void __launch_bounds__(256) k_prior(GDims d, const int32_t* __restrict__ n_nodes,
                                               const RegionStats* __restrict__ st, float* __restrict__ work /*[B,Nmax,2]*/,
                                               const float* __restrict__ contrast, float two_ce2, float* __restrict__ prior) {
    __shared__ float sa[256], sb[256];
    __shared__ double dsum[4][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = n_nodes[b];
    const RegionStats* s = st + (size_t)b * d.Nmax;
    const float* ct = contrast + (size_t)b * d.Nmax;
    float* fg = work + (size_t)b * d.Nmax * 2;
    float* bg = fg + d.Nmax;
    float* pr = prior + (size_t)b * d.Nmax * 3;
    float mn, mx;
    block_minmax(N, [&](int i) { return ct[i]; }, mn, mx, sa, sb);
    for (int i = tid; i < N; i += 256) {
        const float a = s[i].pcy - 0.5f, c = s[i].pcx - 0.5f;
        const float dd = sqrtf(a * a + c * c);
        fg[i] = unit_apply(ct[i], mn, mx) * ggc_expf(-(dd * dd) / two_ce2);
    }
    __syncthreads();
    block_minmax(N, [&](int i) { return fg[i]; }, mn, mx, sa, sb);
    for (int i = tid; i < N; i += 256) fg[i] = unit_apply(fg[i], mn, mx);
    // background colour model from the frame pixels; sums over N in double (order independent to ~1e-16)
    double bs = 0, m0 = 0, m1 = 0, m2 = 0;
    for (int i = tid; i < N; i += 256) {
        const double w = (double)s[i].border;
        bs += w; m0 += w * (double)s[i].mlab[0]; m1 += w * (double)s[i].mlab[1]; m2 += w * (double)s[i].mlab[2];
    }
    dsum[0][tid] = bs; dsum[1][tid] = m0; dsum[2][tid] = m1; dsum[3][tid] = m2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) for (int c = 0; c < 4; ++c) dsum[c][tid] += dsum[c][tid + o];
        __syncthreads();
    }
    const float bsum = (float)dsum[0][0];
    const float mu0 = (float)(dsum[1][0] / dsum[0][0]), mu1 = (float)(dsum[2][0] / dsum[0][0]),
                mu2 = (float)(dsum[3][0] / dsum[0][0]);
    __syncthreads();
    double var = 0;
    for (int i = tid; i < N; i += 256) {
        const float w = s[i].border / bsum;
        const float d0 = s[i].mlab[0] - mu0, d1 = s[i].mlab[1] - mu1, d2 = s[i].mlab[2] - mu2;
        var += (double)((d0 * d0) * w) + (double)((d1 * d1) * w) + (double)((d2 * d2) * w);
    }
    dsum[0][tid] = var;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) dsum[0][tid] += dsum[0][tid + o]; __syncthreads(); }
    const float var_bg = (float)dsum[0][0];
    const double sigma_bg = var_bg > 1e-6 ? (double)sqrtf(var_bg) : sqrt(1e-6);
    const float den = (float)(2.0 * (sigma_bg + 1e-6) * (sigma_bg + 1e-6));
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        float v = 0.0f;
        if (bsum > 0.0f) {
            const float d0 = s[i].mlab[0] - mu0, d1 = s[i].mlab[1] - mu1, d2 = s[i].mlab[2] - mu2;
            const float dd = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
            v = ggc_expf(-(dd * dd) / den);
        }
        float r = (s[i].border / s[i].safe) * 4.0f;
        r = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
        bg[i] = (v != v) ? v : fmaxf(v, r);
    }
    __syncthreads();
    block_minmax(N, [&](int i) { return bg[i]; }, mn, mx, sa, sb);
    for (int i = tid; i < N; i += 256) {
        const float bgv = unit_apply(bg[i], mn, mx), fgv = fg[i];
        pr[3 * i + 0] = fix_nan(fgv);
        pr[3 * i + 1] = fix_nan(bgv);
        pr[3 * i + 2] = fix_nan(1.0f - fabsf(fgv - bgv));
    }
}

// ---- fill: packed outputs
__global__ void __launch_bounds__(256) k_fill_nodes(GDims d, const int32_t* __restrict__ n_nodes,
                                                    const int64_t* __restrict__ node_ptr,
                                                    const float* __restrict__ feat, const float* __restrict__ prior,
                                                    const RegionStats* __restrict__ st, float* __restrict__ x,
                                                    float* __restrict__ cent, float* __restrict__ area) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes[b]) return;
    const int64_t o = node_ptr[b] + i;
    const size_t li = (size_t)b * d.Nmax + i;
    if (x) {
        for (int c = 0; c < 16; ++c) x[o * 19 + c] = feat[li * 16 + c];
        for (int c = 0; c < 3; ++c) x[o * 19 + 16 + c] = prior[li * 3 + c];
    }
    if (cent) { cent[o * 2] = st[li].cy; cent[o * 2 + 1] = st[li].cx; }
    if (area) area[o] = st[li].area;
}

__global__ void __launch_bounds__(256) k_fill_edges(int B, const int64_t* __restrict__ pair_ptr,
                                                    const int64_t* __restrict__ node_ptr, int global_ids, PairOut p,
                                                    const float* __restrict__ flag, int32_t* __restrict__ src,
                                                    int32_t* __restrict__ dst, float* __restrict__ attr) {
    const int b = blockIdx.y;
    const int off = global_ids ? (int)node_ptr[b] : 0;      // PyG Batch collation offsets
    const int64_t beg = pair_ptr[b], end = pair_ptr[b + 1];
    const int64_t q = beg + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= end) return;
    const int64_t np = end - beg;
    const int64_t e0 = 2 * beg + (q - beg), e1 = e0 + np;   // [lo.., hi..] then mirrored (:303-306)
    if (src) { src[e0] = p.lo[q] + off; src[e1] = p.hi[q] + off; }
    if (dst) { dst[e0] = p.hi[q] + off; dst[e1] = p.lo[q] + off; }
    if (attr) {
        const float a[5] = {p.de[q], p.dxy[q], p.shared[q], p.gc[q], flag[q]};
        for (int c = 0; c < 5; ++c) { attr[e0 * 5 + c] = a[c]; attr[e1 * 5 + c] = a[c]; }
    }
}

} // namespace ggc

using namespace ggc;

extern "C" int ggc_graph_count(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const int32_t* segments,
                               const int32_t* n_nodes, const float* lab, const float* hsv, const float* grad,
                               int connectivity, int n_nonlocal, int64_t* node_ptr, int64_t* edge_ptr) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, segments && n_nodes && lab && hsv && grad && node_ptr && edge_ptr, GGC_E_INVALID_ARG, "null pointer");
    GGC_REQUIRE(ctx, connectivity == 4 || connectivity == 8, GGC_E_INVALID_ARG, "connectivity must be 4 or 8");
    GGC_REQUIRE(ctx, n_nonlocal >= 0 && n_nonlocal <= 64, GGC_E_INVALID_ARG, "n_nonlocal out of range");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ctx->graph.valid = false;

    // sync 1: node counts
    std::vector<int32_t> nn(B);
    GGC_HIP(ctx, hipMemcpyAsync(nn.data(), n_nodes, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
    GGC_HIP(ctx, hipStreamSynchronize(st));
    int Nmax = 1;
    GraphState& gs = ctx->graph;
    gs.node_ptr.assign(B + 1, 0);
    for (int b = 0; b < B; ++b) {
        GGC_REQUIRE(ctx, nn[b] >= 1 && (size_t)nn[b] <= (size_t)H * W, GGC_E_SHAPE, "n_nodes[%d]=%d is invalid", b, nn[b]);
        Nmax = std::max(Nmax, nn[b]);
        gs.node_ptr[b + 1] = gs.node_ptr[b] + nn[b];
    }
    GGC_REQUIRE(ctx, (size_t)B * Nmax * Nmax * 4 < (size_t)64 << 30, GGC_E_OOM,
                "dense adjacency of %d x %d^2 exceeds 64 GiB", B, Nmax);
    GDims d{B, H, W, Nmax, connectivity, n_nonlocal};
    const size_t BN = (size_t)B * Nmax;

    int4* bbox = scratch_t<int4>(ctx, S_G_AUX, BN);
    int32_t* border = scratch_t<int32_t>(ctx, S_G_AUX2, BN);
    uint32_t* small = scratch_t<uint32_t>(ctx, S_G_AUX3, (size_t)B * 16);   // gmax[B] | maxima[B,5] | totals[B,3]
    RegionStats* stats = scratch_t<RegionStats>(ctx, S_G_STATS, BN);
    int32_t* dense = scratch_t<int32_t>(ctx, S_G_PAIRCNT, BN * Nmax);
    int32_t* cnt_adj = scratch_t<int32_t>(ctx, S_G_NL, BN * 2);
    float* feat = scratch_t<float>(ctx, S_G_FEAT, BN * 16);
    float* prior = scratch_t<float>(ctx, S_G_PRIOR, BN * 3);
    float* work = scratch_t<float>(ctx, S_G_X, BN * 3);
    if (!bbox || !border || !small || !stats || !dense || !cnt_adj || !feat || !prior || !work) return GGC_E_OOM;
    int32_t* cnt_nl = cnt_adj + BN;
    uint32_t* gmax = small;
    uint32_t* maxima = small + B;
    int32_t* totals = reinterpret_cast<int32_t*>(small + (size_t)B * 6);
    float* contrast = work + BN * 2;

    hipLaunchKernelGGL(k_bbox_init, dim3(cdiv(BN, 256)), dim3(256), 0, st, BN, bbox);
    GGC_HIP(ctx, hipMemsetAsync(border, 0, sizeof(int32_t) * BN, st));
    GGC_HIP(ctx, hipMemsetAsync(small, 0, sizeof(uint32_t) * (size_t)B * 16, st));
    GGC_HIP(ctx, hipMemsetAsync(dense, 0, sizeof(int32_t) * BN * Nmax, st));
    const dim3 pix(cdiv(W, 64), cdiv(H, 4), B);
    hipLaunchKernelGGL(k_bbox, dim3(cdiv(W, 64), cdiv(H, BBOX_ROWS), B), dim3(256), 0, st, d, segments, grad, bbox, border, gmax);
    {
        ProfScope prof(ctx, st, "graph_stats");
        const size_t stats_lds = ((size_t)W * 15 + 2) * sizeof(float);  // 3 label rows, lab, hsv, grad, grad/max, x/W as f32 and f64
        static std::atomic<int> stats_lds_dev[64];                      // largest size set so far, per device
        std::atomic<int>& stats_lds_set = stats_lds_dev[ctx->device & 63];
        if ((int)stats_lds > 48 * 1024 && stats_lds_set.load(std::memory_order_acquire) < (int)stats_lds) {
            GGC_REQUIRE(ctx, stats_lds <= 160 * 1024, GGC_E_UNSUPPORTED, "image width %d too large for the statistics kernel", W);
            GGC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stats), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)stats_lds));
            stats_lds_set.store((int)stats_lds, std::memory_order_release);
        }
        hipLaunchKernelGGL(k_stats, dim3(cdiv(Nmax, 64), B), dim3(64), stats_lds, st, d, segments, n_nodes, lab, hsv, grad,
                           bbox, border, gmax, stats);
    }
    hipLaunchKernelGGL(k_adj, pix, dim3(256), 0, st, d, segments, dense);
    GGC_LAUNCH_CHECK(ctx);
    const dim3 rows(cdiv(Nmax, 4), B);
    if (n_nonlocal > 0) {
        const size_t lds = (size_t)4 * Nmax * sizeof(float);
        GGC_REQUIRE(ctx, lds <= 64 * 1024, GGC_E_UNSUPPORTED, "k-NN row buffer for %d nodes exceeds LDS", Nmax);
        ProfScope prof(ctx, st, "graph_knn");
        hipLaunchKernelGGL(k_knn, rows, dim3(256), lds, st, d, n_nodes, stats, dense);
    }
    hipLaunchKernelGGL(k_rowcount, rows, dim3(256), 0, st, d, n_nodes, dense, cnt_adj, cnt_nl);
    hipLaunchKernelGGL(k_rowscan, dim3(B), dim3(256), 0, st, d, n_nodes, cnt_adj, cnt_nl, totals);
    GGC_LAUNCH_CHECK(ctx);

    // sync 2: pair counts
    std::vector<int32_t> tot((size_t)B * 3);
    GGC_HIP(ctx, hipMemcpyAsync(tot.data(), totals, sizeof(int32_t) * B * 3, hipMemcpyDeviceToHost, st));
    GGC_HIP(ctx, hipStreamSynchronize(st));
    std::vector<int64_t> pair_ptr(B + 1, 0);
    gs.edge_ptr.assign(B + 1, 0);
    for (int b = 0; b < B; ++b) {
        pair_ptr[b + 1] = pair_ptr[b] + tot[3 * b] + tot[3 * b + 1];
        gs.edge_ptr[b + 1] = 2 * pair_ptr[b + 1];
    }
    const int64_t n_pairs = pair_ptr[B];
    int64_t* ptrs = scratch_t<int64_t>(ctx, S_G_PTR, (size_t)2 * (B + 1));
    int32_t* plo = scratch_t<int32_t>(ctx, S_G_PAIRS, (size_t)std::max<int64_t>(n_pairs, 1) * 2);
    float* pattr = scratch_t<float>(ctx, S_G_EDGE_ATTR, (size_t)std::max<int64_t>(n_pairs, 1) * 5);
    if (!ptrs || !plo || !pattr) return GGC_E_OOM;
    GGC_HIP(ctx, hipMemcpyAsync(ptrs, gs.node_ptr.data(), sizeof(int64_t) * (B + 1), hipMemcpyHostToDevice, st));
    GGC_HIP(ctx, hipMemcpyAsync(ptrs + (B + 1), pair_ptr.data(), sizeof(int64_t) * (B + 1), hipMemcpyHostToDevice, st));
    GGC_HIP(ctx, hipStreamSynchronize(st));   // the host vectors above go out of scope
    const size_t np1 = (size_t)std::max<int64_t>(n_pairs, 1);
    PairOut po{plo, plo + np1, pattr, pattr + np1, pattr + 2 * np1, pattr + 3 * np1};
    float* flag = pattr + 4 * np1;
    int64_t max_pairs = 0;
    for (int b = 0; b < B; ++b) max_pairs = std::max(max_pairs, pair_ptr[b + 1] - pair_ptr[b]);
    hipLaunchKernelGGL(k_extract, rows, dim3(256), 0, st, d, n_nodes, dense, stats, cnt_adj, cnt_nl, ptrs + (B + 1),
                       totals, po, maxima);
    if (max_pairs > 0)
        hipLaunchKernelGGL(k_pair_final, dim3(cdiv(max_pairs, 256), B), dim3(256), 0, st, B, ptrs + (B + 1), totals,
                           maxima, po, flag);
    hipLaunchKernelGGL(k_node_feats, dim3(B), dim3(256), 0, st, d, n_nodes, stats, feat);
    {
        ProfScope prof(ctx, st, "graph_prior");
        hipLaunchKernelGGL(k_contrast, rows, dim3(256), 0, st, d, n_nodes, stats, ctx->prior_two_cs2, contrast);
        hipLaunchKernelGGL(k_prior, dim3(B), dim3(256), 0, st, d, n_nodes, stats, work, contrast, ctx->prior_two_ce2, prior);
    }
    GGC_LAUNCH_CHECK(ctx);

    gs.B = B; gs.H = H; gs.W = W;
    gs.n_total = gs.node_ptr[B]; gs.e_total = gs.edge_ptr[B];
    gs.valid = true;
    gs.max_pairs = max_pairs; gs.n_max = Nmax; gs.n_nodes_dev = n_nodes;
    for (int b = 0; b <= B; ++b) { node_ptr[b] = gs.node_ptr[b]; edge_ptr[b] = gs.edge_ptr[b]; }
    return GGC_OK;
}

extern "C" int ggc_graph_fill(ggc_ctx* ctx, ggc_stream stream, float* x, float* centroids, float* area_ratio,
                              int32_t* edge_src, int32_t* edge_dst, float* edge_attr, int global_ids) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GraphState& gs = ctx->graph;
    GGC_REQUIRE(ctx, gs.valid, GGC_E_STATE, "ggc_graph_fill without a preceding successful ggc_graph_count");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int B = gs.B;
    const int64_t max_pairs = gs.max_pairs;
    const int Nmax = gs.n_max;
    const int32_t* n_nodes = gs.n_nodes_dev;
    GDims d{B, gs.H, gs.W, Nmax, 4, 0};
    const int64_t n_pairs = gs.e_total / 2;
    const size_t np1 = (size_t)std::max<int64_t>(n_pairs, 1);
    const int64_t* ptrs = reinterpret_cast<const int64_t*>(ctx->slots[S_G_PTR].p);
    int32_t* plo = reinterpret_cast<int32_t*>(ctx->slots[S_G_PAIRS].p);
    float* pattr = reinterpret_cast<float*>(ctx->slots[S_G_EDGE_ATTR].p);
    PairOut po{plo, plo + np1, pattr, pattr + np1, pattr + 2 * np1, pattr + 3 * np1};
    const float* flag = pattr + 4 * np1;
    hipLaunchKernelGGL(k_fill_nodes, dim3(cdiv(Nmax, 256), B), dim3(256), 0, st, d, n_nodes, ptrs,
                       reinterpret_cast<const float*>(ctx->slots[S_G_FEAT].p),
                       reinterpret_cast<const float*>(ctx->slots[S_G_PRIOR].p),
                       reinterpret_cast<const RegionStats*>(ctx->slots[S_G_STATS].p), x, centroids, area_ratio);
    if (max_pairs > 0 && (edge_src || edge_dst || edge_attr))
        hipLaunchKernelGGL(k_fill_edges, dim3(cdiv(max_pairs, 256), B), dim3(256), 0, st, B, ptrs + (B + 1), ptrs,
                           global_ids, po, flag, edge_src, edge_dst, edge_attr);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

extern "C" int ggc_graph_prior_sigmas(ggc_ctx* ctx, double centre_sigma, double contrast_sigma) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, centre_sigma > 0.0 && contrast_sigma > 0.0, GGC_E_INVALID_ARG, "prior sigmas must be positive");
    ctx->prior_two_ce2 = (float)(2 * centre_sigma * centre_sigma);     // python floats, then float32 operands (graph_builder.py:408,415)
    ctx->prior_two_cs2 = (float)(2 * contrast_sigma * contrast_sigma);
    return GGC_OK;
}
