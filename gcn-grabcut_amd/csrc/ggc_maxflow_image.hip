// ggc_maxflow_image.hip — the max-flow of ggc_maxflow.hip with the round control on the device:
// ONE resident workgroup per image runs every round (global relabel -> push-relabel sweeps -> ... -> no active pixel)
// inside a single launch.  Same algorithm, same tiles, same canonical result (a pixel is foreground iff it cannot
// reach the sink in the final residual graph); what changes is who decides what runs next.
//
// Why.  Driven from the host, a max-flow of a 256-image batch was ~1250 launches and ~110 blocking read-backs of
// list sizes (round 1: 81 % of the GPU time of a step, most of it launch latency in rounds where a handful of
// images still had a few hundred active pixels).  Images are independent, so nothing of that needs a grid-wide
// decision:
//  * a workgroup of 12 waves owns one image (256 images = 256 CUs).  No other workgroup ever touches the image, so
//    there is no inter-workgroup barrier, no flag hand-off between CUs and no co-residency requirement: nothing in
//    here can wait for a workgroup that is not scheduled.
//  * a WAVE is the worker: it takes the next tile of the image's work list (LDS counter), stages it in its private
//    slice of LDS (relabel: 32x32 labels + halo; push: 32x8 excess / 8 residual planes / labels + halo), sweeps it
//    wave-synchronously (no barrier inside a visit) and writes back.  Tiles in flight at once = 12 per image.
//  * work lists are per image: membership bitmap in LDS (one atomicOr per flagged tile), compacted into a list at
//    every pass boundary (popcount + wave scan), one workgroup barrier per pass.
//  * the push schedule (passes per round, tail handling) is decided per image from ITS active-pixel count, so an
//    image in its tail does not wait for the batch and a finished image frees its CU for the next launch.
// Concurrency between the waves of an image is the lock-free push-relabel of ggc_maxflow.hip: pushes across a tile
// edge are global atomics, ring pixels are written back as atomic deltas, stale labels are lower bounds.
//
// Memory visibility.  All sharing is inside one workgroup (one CU, one L1).  Data another wave may change during a
// pass (excess, residual capacities, labels of the halo) is loaded with sc1 (L2-served) loads; every pass boundary
// is a workgroup barrier followed by an agent-scope acquire (L1 invalidate), after which plain loads are fresh.
#include "ggc_gc.h"
#include "ggc_mf_sweep.h"
#include <algorithm>
#include <cstdlib>

namespace ggc {
namespace {

constexpr int MW = 12, MT = MW * 64;                 // waves / threads of an image's workgroup
constexpr int RT = MF_RT, PT_W = MF_PT_W, PT_H = MF_PT_H, PT_N = PT_W * PT_H;
constexpr int MAX_TILES = 8192, BM_WORDS = MAX_TILES / 32;
constexpr int RT_PX = RT * RT / 64;                  // pixels per lane in a relabel tile (16)
constexpr int PT_PX = PT_N / 64;                     // pixels per lane in a push tile (4)
constexpr int RT_HALO = (RT + 2) * (RT + 2), PT_HALO = (PT_H + 2) * (PT_W + 2);

struct Sched { int passes0, passes, inner, tail_active, tail_passes, tail_inner, max_rounds, push_only; };   // push_only: one push phase on the labels it is given

struct PushLds { int ex[PT_N]; int sk[PT_N]; int d[PT_H + 2][PT_W + 2]; int rc[8][PT_N]; unsigned short act[PT_N]; };
struct RelaxLds { int d[RT + 2][RT + 2]; uint32_t m[RT][RT / 4]; int o[RT][RT]; };
union WaveLds { PushLds push; RelaxLds relax; };
struct ImgLds {
    uint32_t bm[BM_WORDS];          // tiles on the NEXT list
    int wsum[4];
    int n_list, head, active, pad;
    WaveLds w[MW];                  // last: a launch with fewer waves (the push-only tail) allocates only its own slices
};
static size_t img_lds_bytes(int waves) { return offsetof(ImgLds, w) + (size_t)waves * sizeof(WaveLds); }

// in-kernel stamps of the trace build (GGC_MF_TRACE): wall_clock64 ticks at 100 MHz
struct VisitProf { long long load = 0, sweep = 0, wb = 0; int sweeps = 0, active = 0; };
#define VP_T() (PROF ? wall_clock64() : 0ll)

__device__ __forceinline__ int ldg(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wave_sync() {        // LDS traffic of one wave is in order: only the compiler needs telling
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void flag_tile(uint32_t* bm, int tile) { atomicOr(&bm[tile >> 5], 1u << (tile & 31)); }
// consecutive lanes mostly flag the same tile: one LDS atomic per run
__device__ __forceinline__ void flag_tile_run(uint32_t* bm, bool flag, int tile, int lane) {
    const int t = flag ? tile : -1;
    const int tp = __shfl_up(t, 1, 64);
    if (flag && (lane == 0 || tp != t)) flag_tile(bm, t);
}
__device__ __forceinline__ int grab(int* head, int lane) {
    int i = 0;
    if (lane == 0) i = atomicAdd(head, 1);
    return __builtin_amdgcn_readfirstlane(i);
}

// bitmap -> list (ascending tile order), bitmap cleared, work counter reset.  Called by every thread after a barrier.
__device__ int compact(ImgLds& L, int n_words, int32_t* __restrict__ list, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    uint32_t w = 0;
    int c = 0, incl = 0;
    if (wv < 4) {
        w = tid < n_words ? L.bm[tid] : 0u;
        c = __popc(w);
        incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) L.wsum[wv] = incl;
    }
    __syncthreads();
    if (wv < 4) {
        int base = incl - c;
        for (int k = 0; k < wv; ++k) base += L.wsum[k];
        while (w) { const int bit = __ffs(w) - 1; list[base++] = tid * 32 + bit; w &= w - 1; }
        if (tid < BM_WORDS) L.bm[tid] = 0u;
    }
    if (tid == 0) { L.n_list = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3]; L.head = 0; }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the previous pass' stores and atomics are in L2: drop this CU's L1 copies
    return L.n_list;
}

// start of a global relabel: d = 1 next to the sink, infinity elsewhere.  The arc masks (rmask bit dir = residual arc
// p -> nb(dir)) are NOT rebuilt from the 8 capacity planes (36 B per pixel and relabel): k_build_graph writes them and
// every push visit keeps them current for the pixels it owns; the few (pixel, direction) pairs another tile can change
// are re-read from the capacities by relax_visit.  Tiles without a pixel away from the sink need no visit.
__device__ void relabel_init(const GcDims& d, const MfTiles& tl, size_t base, const int32_t* __restrict__ snk,
                                                       int32_t* __restrict__ dist, uint32_t* bm, int tid) {
    const int lane = tid & 63;
    constexpr int U = 4;                                   // pixels per thread and trip: U loads in flight
    const int mt = blockDim.x;
    const int qw = mt / d.W, rw = mt % d.W;
    int y = tid / d.W, x = tid % d.W;
    for (int p0 = tid; p0 < d.P; p0 += U * mt) {
        int s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = snk[base + min(p0 + u * mt, d.P - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * mt;
            if (p < d.P) dist[base + p] = s[u] > 0 ? 1 : DINF;
            flag_tile_run(bm, p < d.P && s[u] <= 0, (y / RT) * tl.rt_x + x / RT, lane);
            x += rw; y += qw;
            if (x >= d.W) { x -= d.W; ++y; }
        }
    }
}

typedef RelaxLds RelaxTile;                          // labels + halo, inverted arc masks (1 byte per pixel); sweeps: ggc_mf_sweep.h


// one visit of a 32x32 relabel tile by one wave: relax to the local fixpoint, write back, flag neighbours whose halo changed
template <bool PROF>
__device__ void relax_visit(const GcDims& d, const MfTiles& tl, int tile, size_t base, size_t BP,
                                                      const int32_t* __restrict__ rc, int32_t* __restrict__ dist,
                                                      const uint8_t* __restrict__ rmask, RelaxTile& S, uint32_t* bm, int lane_in, VisitProf& vp) {
    const long long t_a = VP_T();
    // The per-lane index arithmetic below depends only on the lane: left alone, the optimiser hoists all of it out of the
    // kernel's loops to the kernel entry and spills it (scratch reloads inside a visit cost more than recomputing).
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tyi = tile / tl.rt_x, txi = tile % tl.rt_x;
    const int ty0 = tyi * RT, tx0 = txi * RT;
    const int lx = lane & 31, h = lane >> 5;
    int* sd = &S.d[0][0];
    int hv[(RT_HALO + 63) / 64];
#pragma unroll
    for (int k = 0; k < (RT_HALO + 63) / 64; ++k) {
        // every load is issued unconditionally from a clamped address (a load under a branch is waited for on the spot:
        // 19 dependent round trips instead of one)
        const int i = min(lane + k * 64, RT_HALO - 1);
        const int gy = ty0 + i / (RT + 2) - 1, gx = tx0 + i % (RT + 2) - 1;
        hv[k] = ldg(dist + base + (size_t)min(max(gy, 0), d.H - 1) * d.W + min(max(gx, 0), d.W - 1));
    }
    // Inverted arc masks (bit set = no arc), one byte per pixel, staged in LDS; pixels outside the image: all blocked.
    // An arc that LEAVES its pixel's 32x8 push tile may have been re-opened by a push from the neighbouring tile after the
    // owner wrote the mask, so those bits come from the capacities themselves: rows with y % 8 == 0 / 7 (arcs up / down;
    // the lane's V-sweep rows 0, 8 / 7, 15) and columns 0 / 31 (arcs left / right; the lane's H-sweep pixel of that column).
    uint8_t* sm = reinterpret_cast<uint8_t*>(&S.m[0][0]);
    uint32_t mv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gy = ty0 + 16 * h + r, gx = tx0 + lx;
        mv[r] = rmask[base + (size_t)min(gy, d.H - 1) * d.W + min(gx, d.W - 1)];
    }
    int fr[4][3], fc[3];
    {
        const size_t cx = min(tx0 + lx, d.W - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // q: rows 0, 7, 8, 15 of the lane's segment
            const int r = (q >> 1) * 8 + ((q & 1) ? 7 : 0);
            const size_t i = base + (size_t)min(ty0 + 16 * h + r, d.H - 1) * d.W + cx;
#pragma unroll
            for (int t = 0; t < 3; ++t) fr[q][t] = ldg(rc + rc_idx(((q & 1) ? 3 + 2 * t : 2 + 2 * t), i));   // 3,5,7 | 2,4,6
        }
        const size_t i = base + (size_t)min(ty0 + lx, d.H - 1) * d.W + min(tx0 + (h ? 31 : 0), d.W - 1);   // H-sweep row lx
        fc[0] = ldg(rc + rc_idx((h ? 1 : 0), i));
        fc[1] = ldg(rc + rc_idx((h ? 5 : 4), i));
        fc[2] = ldg(rc + rc_idx((h ? 6 : 7), i));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gy = ty0 + 16 * h + r, gx = tx0 + lx;
        uint32_t m = ~mv[r] & 0xffu;
        if ((r & 7) == 0 || (r & 7) == 7) {
            const int q = (r >> 3) * 2 + ((r & 7) ? 1 : 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const uint32_t bit = 1u << (((r & 7) ? 3 : 2) + 2 * t);
                m = (fr[q][t] > 0) ? (m & ~bit) : (m | bit);
            }
        }
        sm[(16 * h + r) * RT + lx] = (gx < d.W && gy < d.H) ? (uint8_t)m : (uint8_t)0xffu;
    }
#pragma unroll
    for (int k = 0; k < (RT_HALO + 63) / 64; ++k) {
        const int i = lane + k * 64;
        const int gy = ty0 + i / (RT + 2) - 1, gx = tx0 + i % (RT + 2) - 1;
        if (i < RT_HALO) sd[i] = (gx >= 0 && gx < d.W && gy >= 0 && gy < d.H) ? hv[k] : DINF;
    }
    wave_sync();
    {
        const int col = h ? 31 : 0;
        uint32_t m = sm[lx * RT + col];
        const uint32_t b0 = 1u << (h ? 1 : 0), b1 = 1u << (h ? 5 : 4), b2 = 1u << (h ? 6 : 7);
        m = (fc[0] > 0) ? (m & ~b0) : (m | b0);
        m = (fc[1] > 0) ? (m & ~b1) : (m | b1);
        m = (fc[2] > 0) ? (m & ~b2) : (m | b2);
        if (ty0 + lx < d.H && tx0 + col < d.W) sm[lx * RT + col] = (uint8_t)m;
    }
    wave_sync();
    uint32_t inv_v[4] = {0u, 0u, 0u, 0u}, inv_h[4];         // V sweep: rows 16h .. 16h+15 of column lx; H sweep: row lx, columns 16h .. 16h+15
#pragma unroll
    for (int r = 0; r < 16; ++r) inv_v[r >> 2] |= (uint32_t)sm[(16 * h + r) * RT + lx] << (8 * (r & 3));
#pragma unroll
    for (int k = 0; k < 4; ++k) inv_h[k] = S.m[lx][4 * h + k];
#pragma unroll
    for (int r = 0; r < 16; ++r) S.o[16 * h + r][lx] = S.d[16 * h + r + 1][lx + 1];     // labels before the visit
    bool settled = false;
    const long long t_b = VP_T();
    // a full sweep that changes nothing has checked every pixel against unchanged neighbours: fixpoint
    for (int it = 0; it < 4 * RT; ++it) {
        if (PROF) ++vp.sweeps;
        const int ch = (it & 1) ? relax_sweep_h(S, inv_h, lx, h) : relax_sweep_v(S, inv_v, lx, h);
        wave_sync();
        if (!__any(ch)) { settled = true; break; }
    }
    const long long t_c = VP_T();
    int nbm = settled ? 0 : 1 << 4;                    // bit (dy + 1) * 3 + (dx + 1); own tile when the sweep cap cut it short
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ly = 16 * h + r;
        const int v = S.d[ly + 1][lx + 1];
        if (v != S.o[ly][lx]) {
            dist[base + (size_t)(ty0 + ly) * d.W + tx0 + lx] = v;
            const int Lf = lx == 0, Rt = lx == RT - 1, U = ly == 0, D = ly == RT - 1;
            nbm |= (U & Lf) | U << 1 | (U & Rt) << 2 | Lf << 3 | Rt << 5 | (D & Lf) << 6 | D << 7 | (D & Rt) << 8;
        }
    }
    nbm = wave_or(nbm);
    if (lane < 9 && (nbm >> lane) & 1) {
        const int ty = tyi + lane / 3 - 1, tx = txi + lane % 3 - 1;
        if (ty >= 0 && ty < tl.rt_y && tx >= 0 && tx < tl.rt_x) flag_tile(bm, ty * tl.rt_x + tx);
    }
    wave_sync();
    if (PROF) { const long long t_d = wall_clock64(); vp.load += t_b - t_a; vp.sweep += t_c - t_b; vp.wb += t_d - t_c; }
}

// active pixel = excess that can still reach the sink; their push tiles form the round's first list
__device__ void scan_active(const GcDims& d, const MfTiles& tl, size_t base, const int32_t* __restrict__ ex,
                                                      const int32_t* __restrict__ dist, uint32_t* bm, int* active, int tid) {
    const int lane = tid & 63;
    constexpr int U = 4;
    const int mt = blockDim.x;
    const int qw = mt / d.W, rw = mt % d.W;
    int y = tid / d.W, x = tid % d.W, n = 0;
    for (int p0 = tid; p0 < d.P; p0 += U * mt) {
        int e[U], dd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int pc = min(p0 + u * mt, d.P - 1); e[u] = ex[base + pc]; dd[u] = dist[base + pc]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool a = p0 + u * mt < d.P && e[u] > 0 && dd[u] < DINF;
            flag_tile_run(bm, a, (y / PT_H) * tl.pt_x + x / PT_W, lane);
            n += a ? 1 : 0;
            x += rw; y += qw;
            if (x >= d.W) { x -= d.W; ++y; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0 && n) atomicAdd(active, n);
}

// ---- push tile visit ----------------------------------------------------------------------------------------------
// One wave, one 32x8 tile (4 pixels per lane for loading and write-back).  A sweep costs the wave its whole instruction
// stream whenever ANY lane has an active pixel, and only ~10 % of a visited tile's pixels are active: so every sweep
// first compacts the active pixels into an LDS list (ballot + mbcnt) and then hands ONE active pixel to each lane.
// The pixel's 8 residual capacities and 8 neighbour labels are read in one batch; the arg-min is branch-free.
template <bool PROF>
__device__ void push_visit(const GcDims& d, const MfTiles& tl, int tile, int inner, size_t base, size_t BP,
                                                     int32_t* __restrict__ rc, int32_t* __restrict__ ex, int32_t* __restrict__ snk,
                                                     int32_t* __restrict__ dist, uint8_t* __restrict__ rmask, PushLds& S, uint32_t* bm,
                                                     int lane_in, VisitProf& vp) {
    const long long t_a = VP_T();
    int lane = lane_in;                                    // (see relax_visit)
    asm volatile("" : "+v"(lane));
    const int tyi = tile / tl.pt_x, txi = tile % tl.pt_x;
    const int lx = lane & 31, r0 = lane >> 5;
    const int x = txi * PT_W + lx;
    int e0[PT_PX], sk0[PT_PX], d0[PT_PX], r0v[PT_PX][8], pp[PT_PX];
    bool inb[PT_PX];
    // all 46 loads of the visit are issued unconditionally from clamped addresses, then masked (see relax_visit)
#pragma unroll
    for (int j = 0; j < PT_PX; ++j) {
        const int y = tyi * PT_H + r0 + 2 * j;
        inb[j] = x < d.W && y < d.H;
        pp[j] = y * d.W + x;
        const int pc = min(y, d.H - 1) * d.W + min(x, d.W - 1);
        e0[j] = ldg(ex + base + pc);
        sk0[j] = snk[base + pc];
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) r0v[j][dir] = ldg(rc + rc_idx(dir, base + pc));
    }
    int* sd = &S.d[0][0];
    int hv[(PT_HALO + 63) / 64];
#pragma unroll
    for (int k = 0; k < (PT_HALO + 63) / 64; ++k) {
        const int i = min(lane + k * 64, PT_HALO - 1);
        const int gy = tyi * PT_H + i / (PT_W + 2) - 1, gx = txi * PT_W + i % (PT_W + 2) - 1;
        hv[k] = ldg(dist + base + (size_t)min(max(gy, 0), d.H - 1) * d.W + min(max(gx, 0), d.W - 1));
    }
#pragma unroll
    for (int j = 0; j < PT_PX; ++j) {
        const int slot = lane + 64 * j;
        if (!inb[j]) { e0[j] = 0; sk0[j] = 0; }
        S.ex[slot] = e0[j];
        S.sk[slot] = sk0[j];
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) {
            if (!inb[j]) r0v[j][dir] = 0;
            S.rc[dir][slot] = r0v[j][dir];
        }
    }
#pragma unroll
    for (int k = 0; k < (PT_HALO + 63) / 64; ++k) {
        const int i = lane + k * 64;
        const int gy = tyi * PT_H + i / (PT_W + 2) - 1, gx = txi * PT_W + i % (PT_W + 2) - 1;
        if (i < PT_HALO) sd[i] = (gx >= 0 && gx < d.W && gy >= 0 && gy < d.H) ? hv[k] : DINF;
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < PT_PX; ++j) d0[j] = S.d[r0 + 2 * j + 1][lx + 1];
    const long long t_b = VP_T();
    for (int it = 0; it < inner; ++it) {
        // ---- compact the active pixels of the tile (slot order)
        int n_act = 0;
#pragma unroll
        for (int j = 0; j < PT_PX; ++j) {
            const int slot = lane + 64 * j;
            const bool a = inb[j] && S.ex[slot] > 0 && S.d[r0 + 2 * j + 1][lx + 1] < d.P;
            const unsigned long long m = __ballot(a);
            if (a) S.act[n_act + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned short)slot;
            n_act += __popcll(m);
        }
        wave_sync();
        if (n_act == 0) break;
        if (PROF) { ++vp.sweeps; vp.active += (lane == 0) ? n_act : 0; }
        // ---- one active pixel per lane
        for (int k0 = 0; k0 < n_act; k0 += 64) {
            const int k = k0 + lane;
            if (k < n_act) {
                const int slot = S.act[k], ly = slot >> 5, plx = slot & 31;
                const int e = __hip_atomic_load(&S.ex[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int dp = S.d[ly + 1][plx + 1];
                const int sk = S.sk[slot];
                int r[8], hq[8];
#pragma unroll
                for (int dir = 0; dir < 8; ++dir) {
                    r[dir] = __hip_atomic_load(&S.rc[dir][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    hq[dir] = S.d[ly + 1 + dir_dy(dir)][plx + 1 + dir_dx(dir)];
                }
                int hmin = sk > 0 ? 0 : DINF, best = sk > 0 ? 8 : -1, rb = 0;
#pragma unroll
                for (int dir = 0; dir < 8; ++dir) {
                    const bool ok = r[dir] > 0 && hq[dir] < hmin;
                    hmin = ok ? hq[dir] : hmin; best = ok ? dir : best; rb = ok ? r[dir] : rb;
                }
                if (best >= 0 && dp > hmin) {
                    if (best == 8) {
                        const int dl = min(e, sk);
                        S.sk[slot] = sk - dl;                          // only this lane touches the pixel's sink link
                        atomicSub(&S.ex[slot], dl);
                    } else {
                        const int dl = min(e, rb);
                        atomicSub(&S.rc[best][slot], dl);
                        atomicSub(&S.ex[slot], dl);
                        const int bx = dir_dx(best), by = dir_dy(best);
                        const int qlx = plx + bx, qly = ly + by;
                        if (qlx >= 0 && qlx < PT_W && qly >= 0 && qly < PT_H) {
                            const int qt = qly * PT_W + qlx;
                            atomicAdd(&S.rc[best ^ 1][qt], dl);
                            atomicAdd(&S.ex[qt], dl);
                        } else {                                        // across the tile edge: straight to global memory
                            const int gy = tyi * PT_H + ly + by, gx = txi * PT_W + plx + bx;
                            const size_t q = base + (size_t)gy * d.W + gx;
                            atomicAdd(&rc[rc_idx((best ^ 1), q)], dl);
                            atomicAdd(&ex[q], dl);
                            flag_tile(bm, (gy / PT_H) * tl.pt_x + gx / PT_W);
                        }
                    }
                } else {
                    S.d[ly + 1][plx + 1] = (best >= 0 && hmin < DINF) ? hmin + 1 : DINF;
                }
            }
            wave_sync();
        }
    }
    const long long t_c = VP_T();
    int left = 0;
#pragma unroll
    for (int j = 0; j < PT_PX; ++j) {
        // only the border ring can receive pushes from other tiles while this wave holds the tile: the interior is a plain store
        const int slot = lane + 64 * j, ly = r0 + 2 * j, p = pp[j];
        const bool ring = lx == 0 || lx == PT_W - 1 || ly == 0 || ly == PT_H - 1;
        const int e1 = S.ex[slot], sk1 = S.sk[slot], d1 = S.d[ly + 1][lx + 1];
        int r1[8];
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) r1[dir] = S.rc[dir][slot];
        if (!inb[j]) continue;
        if (e1 != e0[j]) { if (ring) atomicAdd(&ex[base + p], e1 - e0[j]); else ex[base + p] = e1; }
        int m1 = 0, chg = 0;
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) {
            m1 |= (r1[dir] > 0) ? (1 << dir) : 0;
            if (r1[dir] != r0v[j][dir]) {
                chg = 1;
                if (ring) atomicAdd(&rc[rc_idx(dir, base + p)], r1[dir] - r0v[j][dir]);
                else rc[rc_idx(dir, base + p)] = r1[dir];
            }
        }
        if (chg) rmask[base + p] = (uint8_t)m1;             // (arcs that leave the tile: see relax_visit)
        if (sk1 != sk0[j]) snk[base + p] = sk1;
        if (d1 != d0[j]) dist[base + p] = d1;
        left |= (e1 > 0 && d1 < d.P) ? 1 : 0;
    }
    if (__any(left) && lane == 0) flag_tile(bm, tile);     // still has work
    wave_sync();
    if (PROF) { const long long t_d = wall_clock64(); vp.load += t_b - t_a; vp.sweep += t_c - t_b; vp.wb += t_d - t_c; }
}

template <bool PROF>
__global__ void __launch_bounds__(MT) k_mf_image(GcDims d, MfTiles tl, Sched sc, const int32_t* __restrict__ state,
                                                 int32_t* __restrict__ rc, int32_t* __restrict__ ex, int32_t* __restrict__ snk,
                                                 int32_t* __restrict__ dist, uint8_t* __restrict__ rmask,
                                                 int32_t* __restrict__ lists, int list_stride, int32_t* __restrict__ err_flag,
                                                 int32_t* __restrict__ stats) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    ImgLds& L = *reinterpret_cast<ImgLds*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (state[b]) return;
    const size_t base = (size_t)b * d.P, BP = (size_t)d.B * d.P;
    int32_t* list = lists + (size_t)b * list_stride;
    const int n_rt = tl.rt_x * tl.rt_y, n_pt = tl.pt_x * tl.pt_y;
    const int rt_words = (n_rt + 31) / 32, pt_words = (n_pt + 31) / 32;
    if (tid < BM_WORDS) L.bm[tid] = 0u;
    if (tid == 0) { L.active = 0; L.head = 0; L.n_list = 0; }
    __syncthreads();
    int n_relax_pass = 0, n_push_pass = 0, n_relax_visit = 0, n_push_visit = 0, round = 0;
    bool converged = false;
    long long t_init = 0, t_relax = 0, t_scan = 0, t_push = 0, t0 = 0;
    VisitProf vr, vq;
    const long long t_begin = stats ? wall_clock64() : 0;
#define MFI_TICK() do { if (stats) t0 = wall_clock64(); } while (0)
#define MFI_TOCK(acc) do { if (stats) acc += wall_clock64() - t0; } while (0)
    for (; round < sc.max_rounds; ++round) {
        // ---- global relabel: exact distances to the sink in the residual graph
        MFI_TICK();
        int n = 0;
        if (!sc.push_only) {
            relabel_init(d, tl, base, snk, dist, L.bm, tid);
            __syncthreads();
            n = compact(L, rt_words, list, tid);
        }
        MFI_TOCK(t_init);
        MFI_TICK();
        while (n > 0) {
            ++n_relax_pass;
            for (int i = grab(&L.head, lane); i < n; i = grab(&L.head, lane)) {
                relax_visit<PROF>(d, tl, list[i], base, BP, rc, dist, rmask, L.w[wv].relax, L.bm, lane, vr);
                ++n_relax_visit;
            }
            __syncthreads();
            n = compact(L, rt_words, list, tid);
        }
        MFI_TOCK(t_relax);
        // ---- who still has work?
        MFI_TICK();
        scan_active(d, tl, base, ex, dist, L.bm, &L.active, tid);
        __syncthreads();
        const int active = L.active;
        if (PROF && tid == 0 && round < 8) {
            int32_t* q = stats + (size_t)gridDim.x * 32 + (size_t)b * 16;
            q[round] = (int)(wall_clock64() - t_begin); q[8 + round] = active;
        }
        n = compact(L, pt_words, list, tid);               // (its barriers also order the read of L.active against the reset)
        if (tid == 0) L.active = 0;
        MFI_TOCK(t_scan);
        if (active == 0) { converged = true; break; }
        MFI_TICK();
        // ---- push-relabel sweeps; few active pixels: their labels stay exact, so more (cheap) passes beat another global relabel
        const bool tail = sc.push_only || active <= sc.tail_active;
        const int passes = tail ? sc.tail_passes : (round == 0 ? sc.passes0 : sc.passes);
        const int inner = tail ? sc.tail_inner : sc.inner;
        for (int pass = 0; pass < passes && n > 0; ++pass) {
            ++n_push_pass;
            for (int i = grab(&L.head, lane); i < n; i = grab(&L.head, lane)) {
                push_visit<PROF>(d, tl, list[i], inner, base, BP, rc, ex, snk, dist, rmask, L.w[wv].push, L.bm, lane, vq);
                ++n_push_visit;
            }
            __syncthreads();
            n = compact(L, pt_words, list, tid);
        }
        MFI_TOCK(t_push);
        if (sc.push_only) { converged = true; break; }      // the caller relabels
    }
    if (!converged && tid == 0) atomicOr(err_flag, 1);
    if (stats) {
        // per image: rounds, relabel passes, push passes, then the tile visits of every wave summed
        if (tid == 0) {
            stats[b * 16 + 0] = round; stats[b * 16 + 1] = n_relax_pass; stats[b * 16 + 2] = n_push_pass;
            stats[b * 16 + 5] = (int)(wall_clock64() - t_begin); stats[b * 16 + 6] = (int)t_init; stats[b * 16 + 7] = (int)t_relax;
            stats[b * 16 + 8] = (int)t_scan; stats[b * 16 + 9] = (int)t_push;
        }
        if (lane == 0) { atomicAdd(&stats[b * 16 + 3], n_relax_visit); atomicAdd(&stats[b * 16 + 4], n_push_visit); }
        if (PROF) {
            int32_t* q = stats + (size_t)gridDim.x * 16 + (size_t)b * 16;     // second table: visit phases, summed over the waves
            if (lane == 0) {
                atomicAdd(&q[0], (int)vr.load); atomicAdd(&q[1], (int)vr.sweep); atomicAdd(&q[2], (int)vr.wb); atomicAdd(&q[3], vr.sweeps);
                atomicAdd(&q[4], (int)vq.load); atomicAdd(&q[5], (int)vq.sweep); atomicAdd(&q[6], (int)vq.wb); atomicAdd(&q[7], vq.sweeps);
            }
            if (lane == 0) atomicAdd(&q[8], vq.active);
        }
    }
}

int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::max(1, std::atoi(e)) : dflt;
}

} // namespace

bool maxflow_image_fits(const GcDims& d) {
    return (int64_t)cdiv(d.W, PT_W) * cdiv(d.H, PT_H) <= MAX_TILES && d.W >= 1 && d.H >= 1;
}

int maxflow_image(ggc_ctx* ctx, hipStream_t st, const GcDims& d, const int32_t* state, int32_t* rc, int32_t* ex,
                  int32_t* snk, int32_t* dist, uint8_t* rmask, int32_t* err_flag, int push_passes, int push_inner) {
    if (!maxflow_image_fits(d)) return set_err(ctx, GGC_E_UNSUPPORTED, "image has more push tiles than the per-image kernel's bitmap");
    const MfTiles tl{cdiv(d.W, RT), cdiv(d.H, RT), cdiv(d.W, PT_W), cdiv(d.H, PT_H)};
    const int stride = tl.pt_x * tl.pt_y;                 // >= relabel tiles per image
    int32_t* lists = scratch_t<int32_t>(ctx, S_GC_M, (size_t)d.B * stride);
    if (!lists) return GGC_E_OOM;
    static DeviceOnce attr_done;
    if (attr_done.need(ctx->device)) {
        GGC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mf_image<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)sizeof(ImgLds)));
        GGC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mf_image<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)sizeof(ImgLds)));
        attr_done.done(ctx->device);
    }
    static const Sched sc_full{env_int("GGC_MFI_PASSES0", 12), env_int("GGC_MFI_PASSES", 24), env_int("GGC_MFI_INNER", 8),
                               env_int("GGC_MFI_TAIL_ACTIVE", 256), env_int("GGC_MFI_TAIL_PASSES", 16),
                               env_int("GGC_MFI_TAIL_INNER", 32), 4096, 0};
    Sched sc = sc_full;
    int waves = MW;
    if (push_passes > 0) {
        // fewer waves (4 = 50 KB of LDS) would leave the CU to the other lanes' kernels, but the per-image scans then take
        // three times as long: measured 95 ms per GrabCut stage with 4 waves against 79 with 12 (and 73 without this tail)
        static const int tail_waves = std::min(MW, std::max(4, env_int("GGC_MF_IMAGE_TAIL_WAVES", MW)));
        sc.push_only = 1; sc.tail_passes = push_passes; sc.tail_inner = push_inner; waves = tail_waves;
    }
    int32_t* stats = nullptr;
    static const bool trace = std::getenv("GGC_MF_TRACE") != nullptr;
    if (trace) {
        stats = scratch_t<int32_t>(ctx, S_GC_N, (size_t)d.B * 48);
        if (!stats) return GGC_E_OOM;
        GGC_HIP(ctx, hipMemsetAsync(stats, 0, sizeof(int32_t) * d.B * 48, st));
    }
    {
        ProfScope prof(ctx, st, "maxflow_image");
        if (trace)
            hipLaunchKernelGGL(k_mf_image<true>, dim3(d.B), dim3(waves * 64), img_lds_bytes(waves), st, d, tl, sc, state, rc, ex, snk, dist, rmask,
                               lists, stride, err_flag, stats);
        else
            hipLaunchKernelGGL(k_mf_image<false>, dim3(d.B), dim3(waves * 64), img_lds_bytes(waves), st, d, tl, sc, state, rc, ex, snk, dist, rmask,
                               lists, stride, err_flag, stats);
    }
    GGC_LAUNCH_CHECK(ctx);
    if (trace) {
        std::vector<int32_t> h(d.B * 48);
        GGC_HIP(ctx, hipMemcpyAsync(h.data(), stats, sizeof(int32_t) * d.B * 48, hipMemcpyDeviceToHost, st));
        GGC_HIP(ctx, hipStreamSynchronize(st));
        long long s[10] = {}; int mx[10] = {}; int slow = 0;
        for (int b = 0; b < d.B; ++b) {
            for (int k = 0; k < 10; ++k) { s[k] += h[b * 16 + k]; mx[k] = std::max(mx[k], h[b * 16 + k]); }
            if (h[b * 16 + 5] > h[slow * 16 + 5]) slow = b;
        }
        std::fprintf(stderr, "[ggc maxflow image] %d images: rounds %.1f (max %d), relabel passes %.1f (%d), push passes %.1f (%d), "
                     "relabel visits %.0f (%d), push visits %.0f (%d)\n", d.B, (double)s[0] / d.B, mx[0], (double)s[1] / d.B, mx[1],
                     (double)s[2] / d.B, mx[2], (double)s[3] / d.B, mx[3], (double)s[4] / d.B, mx[4]);
        const double us = 0.01;                           // wall_clock64 ticks at 100 MHz
        std::fprintf(stderr, "    mean image: total %.0f us = init %.0f + relax %.0f + scan %.0f + push %.0f;  slowest image %d: total %.0f us = "
                     "init %.0f + relax %.0f + scan %.0f + push %.0f (rounds %d, relabel visits %d, push visits %d)\n",
                     us * s[5] / d.B, us * s[6] / d.B, us * s[7] / d.B, us * s[8] / d.B, us * s[9] / d.B, slow, us * h[slow * 16 + 5],
                     us * h[slow * 16 + 6], us * h[slow * 16 + 7], us * h[slow * 16 + 8], us * h[slow * 16 + 9], h[slow * 16 + 0],
                     h[slow * 16 + 3], h[slow * 16 + 4]);
        if (const char* path = std::getenv("GGC_MF_TRACE_FILE")) {      // raw per-image table for offline analysis
            if (FILE* f = std::fopen(path, "a")) {
                for (int b = 0; b < d.B; ++b) {
                    std::fprintf(f, "%d %d", b, h[b * 16 + 5]);
                    for (int k = 0; k < 16; ++k) std::fprintf(f, " %d", h[(size_t)d.B * 32 + b * 16 + k]);
                    std::fprintf(f, "\n");
                }
                std::fprintf(f, "#\n");
                std::fclose(f);
            }
        }
        double q[9] = {};
        for (int b = 0; b < d.B; ++b) for (int k = 0; k < 9; ++k) q[k] += h[(size_t)d.B * 16 + b * 16 + k];
        std::fprintf(stderr, "    per relabel visit: load %.2f us, sweeps %.2f us (%.1f sweeps), write-back %.2f us;  per push visit: load %.2f us, "
                     "sweeps %.2f us (%.1f sweeps, %.1f active pixel-sweeps), write-back %.2f us\n",
                     us * q[0] / s[3], us * q[1] / s[3], q[3] / s[3], us * q[2] / s[3], us * q[4] / s[4], us * q[5] / s[4], q[7] / s[4],
                     q[8] / s[4], us * q[6] / s[4]);
    }
    return GGC_OK;
}

} // namespace ggc
