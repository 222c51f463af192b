// ggc_grabcut.hip — C0-C6: GrabCut (reference grabcut.py:81-168, i.e. cv2.grabCut;
// semantics restated from SURVEY.md Appendix A.4, see DESIGN.md "GrabCut").
//
//   init        promotions + degenerate guard (grabcut.py:127-140) or rect mask
//   initGMMs    seeded k-means++ on the uint8 colours (exact integer D^2
//               sampling, 10 assignment steps), GMM fit from exact integer sums
//   calcBeta    exact integer sum of squared neighbour differences
//   n-links     gamma * exp(-beta d^2) quantised to int32 (scale 2^18)
//   x n_iter    assign components | learn GMMs | t-links | max-flow | relabel
//
// Everything that feeds an integer decision is order independent (integer
// atomics) or a fixed IEEE sequence (det_exp / det_log), so the mask equals the
// CPU path bit for bit.
//
// The max-flow itself (lock-free push-relabel on LDS-resident tiles driven by work
// lists, warm-started across the GrabCut iterations) lives in ggc_maxflow.hip.
#include "ggc_gc.h"
#include "ggc_math.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace ggc {

constexpr int NCOMP = 5;
constexpr double CAP_SCALE = 262144.0;   // 2^18
constexpr double GAMMA = 50.0, LAMBDA = 9.0 * GAMMA;
constexpr int CHUNK = 1024;              // pixels per k-means++ sampling chunk


struct Gmm {
    double coef[NCOMP], mean[NCOMP][3], cov[NCOMP][9];
    double inv[NCOMP][3][3], det[NCOMP];
    double isd[NCOMP];                       // 1 / sqrt(det): the per-pixel likelihood multiplies by it (hoisted, same value)
};

// ------------------------------------------------------------------ GMM math
__host__ __device__ inline void gmm_prepare(Gmm& g, int ci, double fix) {   // calcInverseCovAndDeterm
    if (!(g.coef[ci] > 0.0)) return;
    double* c = g.cov[ci];
    double d = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]);
    if (d <= 1e-6 && fix > 0.0) {
        c[0] += fix; c[4] += fix; c[8] += fix;
        d = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]);
    }
    g.det[ci] = d;
    g.isd[ci] = 1.0 / sqrt(d);
    const double id = 1.0 / d;
    g.inv[ci][0][0] = (c[4] * c[8] - c[5] * c[7]) * id;
    g.inv[ci][1][0] = -(c[3] * c[8] - c[5] * c[6]) * id;
    g.inv[ci][2][0] = (c[3] * c[7] - c[4] * c[6]) * id;
    g.inv[ci][0][1] = -(c[1] * c[8] - c[2] * c[7]) * id;
    g.inv[ci][1][1] = (c[0] * c[8] - c[2] * c[6]) * id;
    g.inv[ci][2][1] = -(c[0] * c[7] - c[1] * c[6]) * id;
    g.inv[ci][0][2] = (c[1] * c[5] - c[2] * c[4]) * id;
    g.inv[ci][1][2] = -(c[0] * c[5] - c[2] * c[3]) * id;
    g.inv[ci][2][2] = (c[0] * c[4] - c[1] * c[3]) * id;
}

__device__ __forceinline__ double gmm_comp(const Gmm& g, int ci, const uint8_t* px) {
    if (!(g.coef[ci] > 0.0)) return 0.0;
    const double d0 = (double)px[0] - g.mean[ci][0], d1 = (double)px[1] - g.mean[ci][1], d2 = (double)px[2] - g.mean[ci][2];
    const double mult = d0 * (d0 * g.inv[ci][0][0] + d1 * g.inv[ci][1][0] + d2 * g.inv[ci][2][0])
                      + d1 * (d0 * g.inv[ci][0][1] + d1 * g.inv[ci][1][1] + d2 * g.inv[ci][2][1])
                      + d2 * (d0 * g.inv[ci][0][2] + d1 * g.inv[ci][1][2] + d2 * g.inv[ci][2][2]);
    return g.isd[ci] * det_exp(-0.5 * mult);
}
__device__ __forceinline__ double gmm_total(const Gmm& g, const uint8_t* px) {
    double r = 0.0;
    for (int ci = 0; ci < NCOMP; ++ci) r += g.coef[ci] * gmm_comp(g, ci, px);
    return r;
}
__device__ __forceinline__ int gmm_which(const Gmm& g, const uint8_t* px) {
    int k = 0; double mx = 0.0;
    for (int ci = 0; ci < NCOMP; ++ci) { const double p = gmm_comp(g, ci, px); if (p > mx) { k = ci; mx = p; } }
    return k;
}

__device__ __forceinline__ bool is_bg(uint8_t m) { return m == GGC_BGD || m == GGC_PR_BGD; }

// ------------------------------------------------------------------ mask init
__global__ void __launch_bounds__(256) k_gc_flags(GcDims d, const uint8_t* __restrict__ mask, int32_t* __restrict__ flags) {
    const int b = blockIdx.y;
    int f = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < d.P; p += gridDim.x * blockDim.x) {
        const uint8_t m = mask[(size_t)b * d.P + p];
        f |= (m == GGC_FGD) ? 1 : 0;
        f |= (m == GGC_BGD) ? 2 : 0;
        f |= (m > 3) ? 4 : 0;
        f |= is_bg(m) ? 8 : 16;           // class presence (for initGMMs' non-empty assert)
    }
    for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o, 64);
    // one atomic per block and only for bits the cell does not hold yet: same-address atomics serialise in L2
    __shared__ int s_f[4];
    if ((threadIdx.x & 63) == 0) s_f[threadIdx.x >> 6] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        f = s_f[0] | s_f[1] | s_f[2] | s_f[3];
        if (f & ~__hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&flags[b], f);
    }
}

__global__ void __launch_bounds__(256) k_gc_promote(GcDims d, const int32_t* __restrict__ flags, uint8_t* __restrict__ mask) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    const int f = flags[i / d.P];
    uint8_t m = mask[i];
    if (!(f & 1) && m == GGC_PR_FGD) m = GGC_FGD;
    if (!(f & 2) && m == GGC_PR_BGD) m = GGC_BGD;
    mask[i] = m;
}

__global__ void __launch_bounds__(256) k_gc_rect(GcDims d, const int32_t* __restrict__ rects, uint8_t* __restrict__ mask) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    const int b = (int)(i / d.P), p = (int)(i % d.P);
    const int y = p / d.W, x = p % d.W;
    int x0 = rects[4 * b], y0 = rects[4 * b + 1], w = rects[4 * b + 2], h = rects[4 * b + 3];
    if (x0 < 0) { w += x0; x0 = 0; }
    if (y0 < 0) { h += y0; y0 = 0; }
    if (x0 + w > d.W) w = d.W - x0;
    if (y0 + h > d.H) h = d.H - y0;
    mask[i] = (x >= x0 && x < x0 + w && y >= y0 && y < y0 + h) ? GGC_PR_FGD : GGC_BGD;
}

// state[b]: 0 = run, 1 = degenerate / skip
__global__ void k_gc_state(int B, int mode, const int32_t* __restrict__ f1, const int32_t* __restrict__ f2,
                           int32_t* __restrict__ state, int32_t* __restrict__ err) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int s = 0;
    if ((f1[b] | f2[b]) & 4) { *err = 1; s = 1; }                      // checkMask
    if (mode == 0 && (f2[b] & 3) != 3) s = 1;                          // degenerate trimap (grabcut.py:135-140)
    if (mode != 2 && (f2[b] & 24) != 24) s = 1;                        // a class has no sample
    state[b] = s;
}

// ----------------------------------------------------------- k-means++ (initGMMs)
__device__ __forceinline__ uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct KmState {               // per (image, class)
    uint64_t rng;
    int64_t n;                 // samples in the class
    int32_t K;                 // min(5, n)
    int32_t cen_px[NCOMP];     // pixel index of each k-means++ pick
    double cen[NCOMP][3];
};

// Update D^2 with the newest centre, then per-chunk class counts and D^2 sums.
__global__ void __launch_bounds__(256) k_km_chunks(GcDims d, int k, const uint8_t* __restrict__ img,
                                                   const uint8_t* __restrict__ mask, const int32_t* __restrict__ state,
                                                   const KmState* __restrict__ km, int32_t* __restrict__ d2,
                                                   int64_t* __restrict__ chunk_cnt, int64_t* __restrict__ chunk_d2) {
    __shared__ long long s_cnt[2][4], s_d2[2][4];
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    if (state[b]) return;
    const uint8_t* im = img + (size_t)b * d.P * 3;
    long long cnt[2] = {0, 0}, sd[2] = {0, 0};
    for (int t = 0; t < CHUNK / 256; ++t) {
        const int p = ch * CHUNK + t * 256 + tid;
        if (p >= d.P) continue;
        const int c = is_bg(mask[(size_t)b * d.P + p]) ? 0 : 1;
        int v = 0;
        if (k > 0) {
            const KmState& s = km[b * 2 + c];
            if (k - 1 < s.K) {
                const int q = s.cen_px[k - 1];
                int dd = 0;
                for (int ch3 = 0; ch3 < 3; ++ch3) { const int t3 = (int)im[3 * p + ch3] - (int)im[3 * q + ch3]; dd += t3 * t3; }
                v = (k == 1) ? dd : min(d2[(size_t)b * d.P + p], dd);
                d2[(size_t)b * d.P + p] = v;
            } else v = d2[(size_t)b * d.P + p];
        }
        cnt[c] += 1; sd[c] += v;
    }
    for (int c = 0; c < 2; ++c)
        for (int o = 32; o > 0; o >>= 1) { cnt[c] += __shfl_xor(cnt[c], o, 64); sd[c] += __shfl_xor(sd[c], o, 64); }
    if ((tid & 63) == 0) for (int c = 0; c < 2; ++c) { s_cnt[c][tid >> 6] = cnt[c]; s_d2[c][tid >> 6] = sd[c]; }
    __syncthreads();
    if (tid < 2) {
        const size_t o = ((size_t)b * 2 + tid) * d.n_chunks + ch;
        chunk_cnt[o] = s_cnt[tid][0] + s_cnt[tid][1] + s_cnt[tid][2] + s_cnt[tid][3];
        chunk_d2[o] = s_d2[tid][0] + s_d2[tid][1] + s_d2[tid][2] + s_d2[tid][3];
    }
}

__device__ __forceinline__ long long wave_sum64(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ long long wave_scan64(long long v, int lane) {
    for (int o = 1; o < 64; o <<= 1) {
        const long long u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// One wave per (image, class): draw the k-th centre exactly like the oracle's serial walk (first pixel of the class
// whose running count / D^2 sum exceeds the drawn threshold), as prefix scans over 64 chunks / 64 pixels at a time.
__global__ void __launch_bounds__(128) k_km_pick(GcDims d, int k, uint64_t seed, const uint8_t* __restrict__ img,
                                                 const uint8_t* __restrict__ mask, const int32_t* __restrict__ state,
                                                 const int32_t* __restrict__ d2, const int64_t* __restrict__ chunk_cnt,
                                                 const int64_t* __restrict__ chunk_d2, KmState* __restrict__ km) {
    const int b = blockIdx.x, c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (state[b]) return;
    KmState& s = km[b * 2 + c];
    const int64_t* cc = chunk_cnt + ((size_t)b * 2 + c) * d.n_chunks;
    const int64_t* cd = chunk_d2 + ((size_t)b * 2 + c) * d.n_chunks;
    long long n; int K; uint64_t rng;
    if (k == 0) {
        long long part = 0;
        for (int i = lane; i < d.n_chunks; i += 64) part += cc[i];
        n = wave_sum64(part);
        K = n < NCOMP ? (int)n : NCOMP;
        rng = (seed + (uint64_t)b) * 2 + (uint64_t)c + 1;
        if (lane == 0) { s.n = n; s.K = K; s.rng = rng; }
    } else { n = s.n; K = s.K; rng = s.rng; }
    if (k >= K) return;
    long long total = 0;
    if (k > 0) {
        long long part = 0;
        for (int i = lane; i < d.n_chunks; i += 64) part += cd[i];
        total = wave_sum64(part);
    }
    const uint64_t r = splitmix(rng);
    const bool by_d2 = k > 0 && total > 0;
    const long long t = by_d2 ? (long long)(r % (uint64_t)total) : (long long)(r % (uint64_t)n);
    const int64_t* w = by_d2 ? cd : cc;
    // chunk holding the threshold (the last chunk when no prefix exceeds it)
    long long run = 0;
    int ch = d.n_chunks - 1;
    for (int base = 0; base < d.n_chunks; base += 64) {
        const int i = base + lane;
        const long long v = i < d.n_chunks ? (long long)w[i] : 0;
        const long long incl = wave_scan64(v, lane);
        const unsigned long long hit = __ballot(i < d.n_chunks && run + incl > t);
        if (hit) {
            const int l = __ffsll((long long)hit) - 1;
            ch = base + l;
            run += __shfl(incl - v, l, 64);
            break;
        }
        run += __shfl(incl, 63, 64);
    }
    if (ch == d.n_chunks - 1) {                 // also the fallback: everything before the last chunk
        long long part = 0;
        for (int i = lane; i < d.n_chunks - 1; i += 64) part += w[i];
        run = wave_sum64(part);
    }
    int pick = -1, last = -1;
    const int p_end = min((ch + 1) * CHUNK, d.P);
    for (int base = ch * CHUNK; base < p_end; base += 64) {
        const int p = base + lane;
        const bool mine = p < p_end && (is_bg(mask[(size_t)b * d.P + p]) ? 0 : 1) == c;
        const long long v = mine ? (by_d2 ? (long long)d2[(size_t)b * d.P + p] : 1) : 0;
        const long long incl = wave_scan64(v, lane);
        const unsigned long long any = __ballot(mine);
        if (any) last = base + 63 - __clzll((long long)any);
        const unsigned long long hit = __ballot(mine && run + incl > t);
        if (hit) { pick = base + __ffsll((long long)hit) - 1; break; }
        run += __shfl(incl, 63, 64);
    }
    if (pick < 0) pick = last;                  // cannot happen for consistent sums; keeps the kernel total
    if (lane == 0) {
        s.rng = rng;
        s.cen_px[k] = pick;
        for (int ch3 = 0; ch3 < 3; ++ch3) s.cen[k][ch3] = (double)img[((size_t)b * d.P + pick) * 3 + ch3];
    }
}

// accumulators per (image, class, component): count, 3 sums, 9 products (int64, exact)
constexpr int ACC_W = 13;

// Block-local bins live in LDS as 32-bit counters.  A wave's 64 pixels fall into a handful of (class, component)
// bins, so plain LDS atomics would serialise ~13 deep on one address: every bin slot is kept in BIN_COPIES copies in
// adjacent banks (copy = lane % BIN_COPIES) and summed at the flush.  A block covers BIN_PX pixels per thread, which
// also divides the number of global 64-bit flush atomics.
constexpr int BIN_COPIES = 16, BIN_PX = 4;

__global__ void __launch_bounds__(256) k_km_assign(GcDims d, const uint8_t* __restrict__ img,
                                                   const uint8_t* __restrict__ mask, const int32_t* __restrict__ state,
                                                   const KmState* __restrict__ km, uint8_t* __restrict__ comp,
                                                   unsigned long long* __restrict__ acc) {
    __shared__ unsigned int s_acc[2 * NCOMP * 4 * BIN_COPIES];     // (256 * BIN_PX) x 255 fits 32 bits easily
    const int b = blockIdx.y, tid = threadIdx.x, cp = tid % BIN_COPIES;
    if (state[b]) return;
    for (int i = tid; i < 2 * NCOMP * 4 * BIN_COPIES; i += 256) s_acc[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BIN_PX; ++j) {
        const int p = (blockIdx.x * BIN_PX + j) * 256 + tid;
        if (p < d.P) {
            const size_t gp = (size_t)b * d.P + p;
            const int c = is_bg(mask[gp]) ? 0 : 1;
            const uint8_t* px = img + gp * 3;
            int best = 0; double bd = 0.0;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {                               // class by class: centres through scalar loads
                if (c != cc) continue;
                const KmState& s = km[b * 2 + cc];
                for (int k = 0; k < s.K; ++k) {
                    const double a0 = (double)px[0] - s.cen[k][0], a1 = (double)px[1] - s.cen[k][1], a2 = (double)px[2] - s.cen[k][2];
                    const double dist = (a0 * a0 + a1 * a1) + a2 * a2;
                    if (k == 0 || dist < bd) { best = k; bd = dist; }
                }
            }
            comp[gp] = (uint8_t)best;
            unsigned int* a = s_acc + (c * NCOMP + best) * 4 * BIN_COPIES + cp;
            atomicAdd(&a[0], 1u); atomicAdd(&a[BIN_COPIES], (unsigned int)px[0]);
            atomicAdd(&a[2 * BIN_COPIES], (unsigned int)px[1]); atomicAdd(&a[3 * BIN_COPIES], (unsigned int)px[2]);
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * NCOMP * 4; i += 256) {
        unsigned int v = 0;
#pragma unroll
        for (int k = 0; k < BIN_COPIES; ++k) v += s_acc[i * BIN_COPIES + k];
        if (v) atomicAdd(&acc[((size_t)b * 2 * NCOMP + i / 4) * ACC_W + (i % 4)], (unsigned long long)v);
    }
}

__global__ void k_km_update(int B, const int32_t* __restrict__ state, KmState* __restrict__ km,
                            unsigned long long* __restrict__ acc, int update) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 * NCOMP) return;
    const int b = i / (2 * NCOMP), c = (i / NCOMP) % 2, k = i % NCOMP;
    unsigned long long* a = acc + (size_t)i * ACC_W;
    if (!state[b] && update && a[0] > 0)
        for (int ch = 0; ch < 3; ++ch) km[b * 2 + c].cen[k][ch] = (double)(long long)a[1 + ch] / (double)(long long)a[0];
    for (int j = 0; j < ACC_W; ++j) a[j] = 0;
}

// --------------------------------------------------------------- GMM learning
// MODE 0: components already in comp[] (after k-means). MODE 1: assignGMMsComponents first.
template <int MODE>
__global__ void __launch_bounds__(256) k_gmm_accum(GcDims d, const uint8_t* __restrict__ img,
                                                   const uint8_t* __restrict__ mask, const int32_t* __restrict__ state,
                                                   const Gmm* __restrict__ gmm, uint8_t* __restrict__ comp,
                                                   unsigned long long* __restrict__ acc) {
    // Block-local bins in LDS as 32-bit counters (256 * BIN_PX pixels x 255^2 < 2^32), products stored once per
    // symmetric pair: 10 LDS atomics per pixel into BIN_COPIES-fold privatised slots (see k_km_assign); the exact
    // 64-bit totals are formed by the global flush.
    constexpr int LW = 10;   // count | 3 sums | xx xy xz yy yz zz
    __shared__ unsigned int s_acc[2 * NCOMP * LW * BIN_COPIES];
    const int b = blockIdx.y, tid = threadIdx.x, cp = tid % BIN_COPIES;
    if (state[b]) return;
    for (int i = tid; i < 2 * NCOMP * LW * BIN_COPIES; i += 256) s_acc[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BIN_PX; ++j) {
        const int p = (blockIdx.x * BIN_PX + j) * 256 + tid;
        if (p < d.P) {
            const size_t gp = (size_t)b * d.P + p;
            const int c = is_bg(mask[gp]) ? 0 : 1;
            const uint8_t* px = img + gp * 3;
            int ci = 0;
            if (MODE == 1) {   // class by class: the GMM address stays wave-uniform (scalar loads), waves are mostly one class
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) if (c == cc) ci = gmm_which(gmm[b * 2 + cc], px);
                comp[gp] = (uint8_t)ci;
            } else ci = comp[gp];
            unsigned int* a = s_acc + (c * NCOMP + ci) * LW * BIN_COPIES + cp;
            const unsigned int v0 = px[0], v1 = px[1], v2 = px[2];
            atomicAdd(&a[0], 1u);
            atomicAdd(&a[1 * BIN_COPIES], v0); atomicAdd(&a[2 * BIN_COPIES], v1); atomicAdd(&a[3 * BIN_COPIES], v2);
            atomicAdd(&a[4 * BIN_COPIES], v0 * v0); atomicAdd(&a[5 * BIN_COPIES], v0 * v1); atomicAdd(&a[6 * BIN_COPIES], v0 * v2);
            atomicAdd(&a[7 * BIN_COPIES], v1 * v1); atomicAdd(&a[8 * BIN_COPIES], v1 * v2); atomicAdd(&a[9 * BIN_COPIES], v2 * v2);
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * NCOMP * ACC_W; i += 256) {
        const int bin = i / ACC_W, slot = i % ACC_W;
        int ls = slot;                                   // slots 0..3 map to themselves
        if (slot >= 4) {                                 // 4 + 3 r + c  ->  symmetric pair index
            const int r = (slot - 4) / 3, cc = (slot - 4) % 3;
            const int lo = r < cc ? r : cc, hi = r < cc ? cc : r;
            ls = 4 + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5));
        }
        unsigned int v = 0;
#pragma unroll
        for (int k = 0; k < BIN_COPIES; ++k) v += s_acc[(bin * LW + ls) * BIN_COPIES + k];
        if (v) atomicAdd(&acc[(size_t)b * 2 * NCOMP * ACC_W + i], (unsigned long long)v);
    }
}

// endLearning: one thread per (image, class)
__global__ void k_gmm_learn(int B, const int32_t* __restrict__ state, unsigned long long* __restrict__ acc,
                            Gmm* __restrict__ gmm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2) return;
    unsigned long long* a = acc + (size_t)i * NCOMP * ACC_W;
    if (!state[i / 2]) {
        Gmm& g = gmm[i];
        long long total = 0;
        for (int ci = 0; ci < NCOMP; ++ci) total += (long long)a[ci * ACC_W];
        for (int ci = 0; ci < NCOMP; ++ci) {
            const long long n = (long long)a[ci * ACC_W];
            if (n == 0) { g.coef[ci] = 0.0; continue; }
            const double inv_n = 1.0 / (double)n;
            g.coef[ci] = (double)n / (double)total;
            for (int r = 0; r < 3; ++r) g.mean[ci][r] = (double)(long long)a[ci * ACC_W + 1 + r] * inv_n;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    g.cov[ci][3 * r + c] = (double)(long long)a[ci * ACC_W + 4 + 3 * r + c] * inv_n - g.mean[ci][r] * g.mean[ci][c];
            gmm_prepare(g, ci, 0.01);
        }
    }
    for (int j = 0; j < NCOMP * ACC_W; ++j) a[j] = 0;
}

__global__ void k_gmm_from_model(int B, const double* __restrict__ bgd, const double* __restrict__ fgd, Gmm* __restrict__ gmm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2) return;
    const double* m = (i & 1 ? fgd : bgd) + (size_t)(i / 2) * 65;
    Gmm& g = gmm[i];
    for (int k = 0; k < NCOMP; ++k) g.coef[k] = m[k];
    for (int k = 0; k < 15; ++k) g.mean[k / 3][k % 3] = m[NCOMP + k];
    for (int k = 0; k < 45; ++k) g.cov[k / 9][k % 9] = m[4 * NCOMP + k];
    for (int ci = 0; ci < NCOMP; ++ci) gmm_prepare(g, ci, 0.0);
}

__global__ void k_gmm_to_model(int B, const int32_t* __restrict__ state, const Gmm* __restrict__ gmm,
                               double* __restrict__ bgd, double* __restrict__ fgd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 || state[i / 2]) return;
    double* m = (i & 1 ? fgd : bgd) + (size_t)(i / 2) * 65;
    const Gmm& g = gmm[i];
    for (int k = 0; k < NCOMP; ++k) m[k] = g.coef[k];
    for (int k = 0; k < 15; ++k) m[NCOMP + k] = g.mean[k / 3][k % 3];
    for (int k = 0; k < 45; ++k) m[4 * NCOMP + k] = g.cov[k / 9][k % 9];
}

// ------------------------------------------------------------ beta, n-links, t-links
// plane k: 0 left, 1 up-left, 2 up, 3 up-right
__device__ __forceinline__ bool plane_nb(const GcDims& d, int y, int x, int k, int& q) {
    const int dy = (k == 0) ? 0 : -1, dx = (k == 0 || k == 1) ? -1 : (k == 2 ? 0 : 1);
    const int yy = y + dy, xx = x + dx;
    if (xx < 0 || xx >= d.W || yy < 0) return false;
    q = yy * d.W + xx;
    return true;
}
__device__ __forceinline__ int colour_d2(const uint8_t* a, const uint8_t* b) {
    const int t0 = (int)a[0] - (int)b[0], t1 = (int)a[1] - (int)b[1], t2 = (int)a[2] - (int)b[2];
    return t0 * t0 + t1 * t1 + t2 * t2;
}

__global__ void __launch_bounds__(256) k_beta(GcDims d, const uint8_t* __restrict__ img, unsigned long long* __restrict__ bsum) {
    const int b = blockIdx.y;
    const uint8_t* im = img + (size_t)b * d.P * 3;
    unsigned long long s = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < d.P; p += gridDim.x * blockDim.x) {
        const int y = p / d.W, x = p % d.W;
        for (int k = 0; k < 4; ++k) { int q; if (plane_nb(d, y, x, k, q)) s += (unsigned long long)colour_d2(im + 3 * p, im + 3 * q); }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&bsum[b], s);
}

__global__ void __launch_bounds__(256) k_nweights(GcDims d, const uint8_t* __restrict__ img,
                                                  const unsigned long long* __restrict__ bsum, int32_t* __restrict__ nw) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P) return;
    const int b = (int)(i / d.P), p = (int)(i % d.P);
    const int y = p / d.W, x = p % d.W;
    const uint8_t* im = img + (size_t)b * d.P * 3;
    const long long bs = (long long)bsum[b];
    double beta = 0.0;
    if (bs > 0) beta = 1.0 / (2.0 * (double)bs / (double)(4ll * d.W * d.H - 3ll * d.W - 3ll * d.H + 2));
    const double gdiv = GAMMA / sqrt(2.0);
    for (int k = 0; k < 4; ++k) {
        int q; int32_t w = 0;
        if (plane_nb(d, y, x, k, q)) {
            const double wt = ((k & 1) ? gdiv : GAMMA) * det_exp(-beta * (double)colour_d2(im + 3 * p, im + 3 * q));
            w = (int32_t)rint(wt * CAP_SCALE);
        }
        nw[((size_t)k * d.B + b) * d.P + p] = w;
    }
}

// constructGCGraph: t-links cancelled against each other, clamped to +-lambda; residual arcs from the n-link planes
__global__ void __launch_bounds__(256) k_build_graph(GcDims d, const uint8_t* __restrict__ img,
                                                     const uint8_t* __restrict__ mask, const int32_t* __restrict__ state,
                                                     const Gmm* __restrict__ gmm, const int32_t* __restrict__ nw,
                                                     int32_t* __restrict__ rc, int32_t* __restrict__ ex,
                                                     int32_t* __restrict__ snk, uint8_t* __restrict__ rmask, int warm) {
    // one image per grid row: the image index is uniform, so the two GMMs (140 doubles) come through scalar loads
    const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t BP = (size_t)d.B * d.P;
    if (p >= d.P || state[b]) return;
    const size_t i = (size_t)b * d.P + p;
    const int y = p / d.W, x = p % d.W;
    const uint8_t m = mask[i];
    // Warm start, definite pixel: its t-link (+-lambda) is what it was, so the new balance tw + inflow equals the old one —
    // and a finished solve leaves no pixel with both excess and sink capacity, so excess, sink link, capacities and arc
    // mask are already what this kernel would write (59 % of the bench's pixels are definite background).
    if (warm && (m == GGC_BGD || m == GGC_FGD)) return;
    double dv;
    if (m == GGC_BGD) dv = -LAMBDA;
    else if (m == GGC_FGD) dv = LAMBDA;
    else {
        const uint8_t* px = img + i * 3;
        const double from_src = -det_log(gmm_total(gmm[b * 2 + 0], px));
        const double to_snk = -det_log(gmm_total(gmm[b * 2 + 1], px));
        dv = from_src - to_snk;
        if (dv != dv) dv = 0.0;
        if (dv > LAMBDA) dv = LAMBDA;
        if (dv < -LAMBDA) dv = -LAMBDA;
    }
    const int32_t tw = (int32_t)rint(dv * CAP_SCALE);
    const int32_t* nwb = nw + (size_t)b * d.P;
    // own planes give the arcs towards left / up-left / up / up-right; the mirrored arcs read the neighbour's plane
    const int dirs[4] = {0, 4, 2, 6};
    int32_t inflow = 0;
    int arcs = 0;                                        // bit dir = residual arc towards dir (ggc_maxflow_image.hip keeps it current)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dir = dirs[k];
        const int32_t c0 = dir_nb(d, y, x, dir) >= 0 ? nwb[(size_t)k * BP + p] : 0;
        const int q = dir_nb(d, y, x, dir ^ 1);
        const int32_t c1 = q >= 0 ? nwb[(size_t)k * BP + q] : 0;
        if (warm) {   // keep the n-link flow of the previous iteration: net inflow = sum (residual - capacity)
            const int32_t ra = rc[rc_idx(dir, i)], rb = rc[rc_idx((dir ^ 1), i)];
            inflow += (ra - c0) + (rb - c1);
            arcs |= (ra > 0 ? 1 << dir : 0) | (rb > 0 ? 1 << (dir ^ 1) : 0);
        } else {
            rc[rc_idx(dir, i)] = c0;
            rc[rc_idx((dir ^ 1), i)] = c1;
            arcs |= (c0 > 0 ? 1 << dir : 0) | (c1 > 0 ? 1 << (dir ^ 1) : 0);
        }
    }
    rmask[i] = (uint8_t)arcs;
    // Warm start (dynamic graph cuts): only the t-links change between GrabCut iterations, and adding a
    // constant to both t-links of a pixel never changes the cut, so the old n-link flow stays a valid
    // preflow: the pixel's new terminal balance is its t-link difference plus what its neighbours sent it.
    const int32_t bal = tw + inflow;
    ex[i] = bal > 0 ? bal : 0;
    snk[i] = bal < 0 ? -bal : 0;
}

// estimateSegmentation: probable pixels take the side of the cut; foreground = cannot reach the sink
__global__ void __launch_bounds__(256) k_gc_relabel(GcDims d, const int32_t* __restrict__ state,
                                                    const int32_t* __restrict__ dist, uint8_t* __restrict__ mask) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)d.B * d.P || state[i / d.P]) return;
    const uint8_t m = mask[i];
    if (m == GGC_PR_BGD || m == GGC_PR_FGD) mask[i] = dist[i] >= DINF ? GGC_PR_FGD : GGC_PR_BGD;
}

__global__ void __launch_bounds__(256) k_gc_binary(size_t n, const uint8_t* __restrict__ mask, uint8_t* __restrict__ binary) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) binary[i] = (mask[i] == GGC_FGD || mask[i] == GGC_PR_FGD) ? 1 : 0;
}

} // namespace ggc

using namespace ggc;

extern "C" int ggc_grabcut(ggc_ctx* ctx, ggc_stream stream, int B, int H, int W, const uint8_t* image,
                           uint8_t* mask, const int32_t* rects, double* bgd_model, double* fgd_model,
                           int n_iter, int mode, uint64_t seed, uint8_t* binary) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, B >= 1 && H >= 1 && W >= 1 && B <= 65535, GGC_E_SHAPE, "bad shape B=%d H=%d W=%d", B, H, W);
    GGC_REQUIRE(ctx, (size_t)H * W < (1u << 28), GGC_E_SHAPE, "image too large");
    GGC_REQUIRE(ctx, image && mask, GGC_E_INVALID_ARG, "null pointer");
    GGC_REQUIRE(ctx, mode >= 0 && mode <= 2, GGC_E_INVALID_ARG, "mode must be 0 (mask), 1 (rect) or 2 (eval)");
    GGC_REQUIRE(ctx, mode != 1 || rects, GGC_E_INVALID_ARG, "mode 1 needs rects");
    GGC_REQUIRE(ctx, mode != 2 || (bgd_model && fgd_model), GGC_E_INVALID_ARG, "mode 2 needs both models");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GcDims d{B, H, W, H * W, cdiv((size_t)H * W, CHUNK)};
    const size_t BP = (size_t)B * d.P;

    int32_t* small = scratch_t<int32_t>(ctx, S_GC_A, (size_t)B * 8 + 32);   // f1 | f2 | state | - | max-flow flags [B+8] | err | open lists [2B]
    Gmm* gmm = scratch_t<Gmm>(ctx, S_GC_B, (size_t)B * 2);
    unsigned long long* acc = scratch_t<unsigned long long>(ctx, S_GC_C, (size_t)B * 2 * NCOMP * ACC_W + B);
    uint8_t* comp = scratch_t<uint8_t>(ctx, S_GC_D, BP);
    int32_t* nw = scratch_t<int32_t>(ctx, S_GC_E, BP * 4);
    int32_t* rc = scratch_t<int32_t>(ctx, S_GC_F, BP * 8);
    int32_t* ex = scratch_t<int32_t>(ctx, S_GC_G, BP * 3);
    uint8_t* rmask = scratch_t<uint8_t>(ctx, S_GC_L, BP);
    if (!small || !gmm || !acc || !comp || !nw || !rc || !ex || !rmask) return GGC_E_OOM;
    int32_t* snk = ex + BP;
    int32_t* dist = ex + 2 * BP;
    int32_t *f1 = small, *f2 = small + B, *state = small + 2 * B, *mf_flags = small + 4 * B;
    int32_t* err = small + 6 * B + 8;
    unsigned long long* bsum = acc + (size_t)B * 2 * NCOMP * ACC_W;
    GGC_HIP(ctx, hipMemsetAsync(small, 0, sizeof(int32_t) * ((size_t)B * 8 + 32), st));
    GGC_HIP(ctx, hipMemsetAsync(acc, 0, sizeof(unsigned long long) * ((size_t)B * 2 * NCOMP * ACC_W + B), st));

    if (mode == 1) {
        int32_t* drects = scratch_t<int32_t>(ctx, S_GC_H, (size_t)B * 4);
        if (!drects) return GGC_E_OOM;
        GGC_HIP(ctx, hipMemcpyAsync(drects, rects, sizeof(int32_t) * B * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_gc_rect, dim3(cdiv(BP, 256)), dim3(256), 0, st, d, drects, mask);
    }
    const dim3 red(std::min(cdiv(d.P, 256 * 4), 128), B);
    hipLaunchKernelGGL(k_gc_flags, red, dim3(256), 0, st, d, mask, f1);
    if (mode == 0) hipLaunchKernelGGL(k_gc_promote, dim3(cdiv(BP, 256)), dim3(256), 0, st, d, f1, mask);
    hipLaunchKernelGGL(k_gc_flags, red, dim3(256), 0, st, d, mask, f2);
    hipLaunchKernelGGL(k_gc_state, dim3(cdiv(B, 256)), dim3(256), 0, st, B, mode, f1, f2, state, err);
    GGC_LAUNCH_CHECK(ctx);

    if (mode == 2) {
        hipLaunchKernelGGL(k_gmm_from_model, dim3(cdiv(B * 2, 64)), dim3(64), 0, st, B, bgd_model, fgd_model, gmm);
    } else {
        ProfScope prof(ctx, st, "grabcut_init_gmm");
        KmState* km = scratch_t<KmState>(ctx, S_GC_I, (size_t)B * 2);
        int32_t* d2 = scratch_t<int32_t>(ctx, S_GC_J, BP);
        int64_t* chunks = scratch_t<int64_t>(ctx, S_GC_K, (size_t)B * 2 * d.n_chunks * 2);
        if (!km || !d2 || !chunks) return GGC_E_OOM;
        int64_t* chunk_cnt = chunks;
        int64_t* chunk_d2 = chunks + (size_t)B * 2 * d.n_chunks;
        for (int k = 0; k < NCOMP; ++k) {
            hipLaunchKernelGGL(k_km_chunks, dim3(d.n_chunks, B), dim3(256), 0, st, d, k, image, mask, state, km, d2, chunk_cnt, chunk_d2);
            hipLaunchKernelGGL(k_km_pick, dim3(B), dim3(128), 0, st, d, k, (uint64_t)seed, image, mask, state, d2, chunk_cnt, chunk_d2, km);
        }
        for (int it = 0; it < 10; ++it) {
            hipLaunchKernelGGL(k_km_assign, dim3(cdiv(d.P, 256 * BIN_PX), B), dim3(256), 0, st, d, image, mask, state, km, comp, acc);
            hipLaunchKernelGGL(k_km_update, dim3(cdiv(B * 2 * NCOMP, 256)), dim3(256), 0, st, B, state, km, acc, it < 9 ? 1 : 0);
        }
        hipLaunchKernelGGL((k_gmm_accum<0>), dim3(cdiv(d.P, 256 * BIN_PX), B), dim3(256), 0, st, d, image, mask, state, gmm, comp, acc);
        hipLaunchKernelGGL(k_gmm_learn, dim3(cdiv(B * 2, 64)), dim3(64), 0, st, B, state, acc, gmm);
    }
    GGC_LAUNCH_CHECK(ctx);

    if (n_iter > 0) {
        hipLaunchKernelGGL(k_beta, red, dim3(256), 0, st, d, image, bsum);
        hipLaunchKernelGGL(k_nweights, dim3(cdiv(BP, 256)), dim3(256), 0, st, d, image, bsum, nw);
        GGC_LAUNCH_CHECK(ctx);
        for (int it = 0; it < n_iter; ++it) {
            const bool warm = knobs().mf_warm && it > 0;
            {
                ProfScope prof(ctx, st, "grabcut_gmm");
                hipLaunchKernelGGL((k_gmm_accum<1>), dim3(cdiv(d.P, 256 * BIN_PX), B), dim3(256), 0, st, d, image, mask, state, gmm, comp, acc);
                hipLaunchKernelGGL(k_gmm_learn, dim3(cdiv(B * 2, 64)), dim3(64), 0, st, B, state, acc, gmm);
                hipLaunchKernelGGL(k_build_graph, dim3(cdiv(d.P, 256), B), dim3(256), 0, st, d, image, mask, state, gmm, nw, rc, ex, snk, rmask, warm ? 1 : 0);
            }
            GGC_LAUNCH_CHECK(ctx);
            // who drives the rounds of the max-flow was measured in round 2 ("max-flow drivers"): launches over
            // work lists for the dense phases, one asynchronous launch for each sparse phase (ggc_maxflow_async.hip)
            const int rcode = maxflow(ctx, st, d, state, rc, ex, snk, dist, rmask, small + 6 * B + 16, mf_flags, err + 1, !warm);
            if (rcode) return rcode;
            hipLaunchKernelGGL(k_gc_relabel, dim3(cdiv(BP, 256)), dim3(256), 0, st, d, state, dist, mask);
            GGC_LAUNCH_CHECK(ctx);
        }
    }
    if (bgd_model && fgd_model)
        hipLaunchKernelGGL(k_gmm_to_model, dim3(cdiv(B * 2, 64)), dim3(64), 0, st, B, state, gmm, bgd_model, fgd_model);
    if (binary) hipLaunchKernelGGL(k_gc_binary, dim3(cdiv(BP, 256)), dim3(256), 0, st, BP, mask, binary);
    GGC_LAUNCH_CHECK(ctx);
    std::vector<int32_t> herr;
    int rcode = read_i32(ctx, st, err, 2, herr);
    if (rcode) return rcode;
    GGC_REQUIRE(ctx, herr[0] == 0, GGC_E_INVALID_ARG, "mask holds values outside {0,1,2,3}");
    GGC_REQUIRE(ctx, herr[1] == 0, GGC_E_DEVICE, "max-flow did not converge (code %d)", herr[1]);
    return GGC_OK;
}
