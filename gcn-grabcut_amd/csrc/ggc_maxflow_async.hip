// ggc_maxflow_async.hip — the SPARSE phases of the max-flow of ggc_maxflow.hip as one launch each.
//
// Driven by the host, a sparse phase is a chain of dependent launches: the front of a global relabel crosses one
// tile per launch (12-56 launches per relabel), a unit of excess crosses one tile per push launch (24-64 launches per
// round), and every launch costs a tile visit plus a kernel boundary (~18 us) however little there is to do.  The dense
// phases (the first relabel launches, the first push round) are bandwidth work and stay launches over work lists.
//
// Here the chain runs inside ONE launch, asynchronously: a pool of waves takes tiles from a queue, and a visit that
// changes a neighbour's halo (relabel) or hands excess to a neighbour (push) appends that neighbour to the SAME queue.
// There are no passes, no barriers and no per-image phases: a front advances at the latency of a tile visit by one wave
// (a few microseconds).  A wave never waits for anything but a queue entry or a tile lock held by a RUNNING wave, so
// the launch finishes with any number of resident workgroups.
//
//  * queue: a ticket ring.  A consumer takes ticket t = head++ and waits for ring[t % cap] to carry tag t + 1; a producer
//    takes t = tail++ and publishes {t + 1, payload} into the same slot once it is empty.  `pending` counts entries
//    queued or in flight; the visit that brings it to zero raises `done` and every waiting wave leaves.  A tile is in
//    the queue at most once (membership flag, test-and-set), so `cap` = number of tiles never overflows.
//  * memory: the per-XCD L2s are not coherent with each other inside a launch.  Every word another wave may change —
//    labels, excess, residual capacities, sink links, flags, locks, the ring — is therefore WRITTEN with a device-scope
//    atomic read-modify-write (performed at memory) and READ with an sc1 load; tools/micro/xcd_atomics.hip measured 0
//    stale reads of 1.6e8 for that pairing on MI355X, same XCD and across XCDs.  Kernel boundaries on either side make
//    the plain stores of the dense launches visible.
//  * relabel: labels only fall (atomicMin), stale halos only delay — the neighbour that lowers a halo pixel afterwards
//    re-queues the tile.  The launch ends at the exact fixpoint, so the host reads nothing back.
//  * push: a tile has ONE state word {queued, busy}.  A wave that takes a tile marks it busy and sweeps its private LDS
//    copy (two waves on one tile would push the same excess twice); a neighbour that hands excess over meanwhile only sets
//    `queued`, and the holder re-queues the tile when it lets go.  Everything goes back as atomic deltas, exactly like the
//    border ring of the launch-driven kernel.  A queue entry carries its hop count: chains stop after `gen_max` hops (the
//    equivalent of the launch count of a host-driven round); what is left is picked up by the next global relabel + active
//    scan, as before.
//  * a hop costs memory round trips (~2 us each for device-scope atomics and sc1 loads), so both kernels FOLLOW: the wave
//    that finishes a visit continues with one of the tiles it would have queued (no ring traffic on the critical path), and
//    the push works on 32x16 / 32x32 tiles with an active-pixel bitmask, where a sweep costs what the active pixels cost
//    and one visit carries excess across the whole tile.
#include "ggc_gc.h"
#include "ggc_mf_sweep.h"
#include <algorithm>
#include <cstdlib>

namespace ggc {
namespace {

constexpr int RT = MF_RT, PT_W = MF_PT_W, PT_H = MF_PT_H, PT_N = PT_W * PT_H;
constexpr int PT_PX = PT_N / 64;                     // pixels per lane in a push tile (4)
constexpr int RT_HALO = (RT + 2) * (RT + 2), PT_HALO = (PT_H + 2) * (PT_W + 2);
constexpr long long AQ_TIMEOUT = 200000000ll;        // 2 s of wall_clock64 (100 MHz): a wave that waits longer reports and ends the launch

__device__ __forceinline__ int ldg(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ldg64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // this wave's loads, stores and atomics are done

// consumer side, lane 0 of a wave; returns the payload (>= 0) to every lane, or -1 when the launch is over
__device__ __forceinline__ int aq_pop(unsigned long long* __restrict__ ring, int32_t* __restrict__ q, int cap, int lane,
                                      int32_t* __restrict__ err_flag) {
    int payload = -1;
    if (lane == 0) {
        const unsigned t = (unsigned)atomicAdd(&q[AQ_HEAD], 1);
        unsigned long long* slot = ring + t % (unsigned)cap;
        const long long t0 = wall_clock64();
        for (int n = 0;; ++n) {
            const unsigned long long v = ldg64(slot);
            if ((unsigned)(v >> 32) == t + 1u) { atomicExch(slot, 0ull); payload = (int)(unsigned)v; break; }
            if ((n & 3) == 3) {
                if (ldg(&q[AQ_DONE]) || ldg(&q[AQ_PENDING]) == 0) break;       // nothing queued, nothing in flight
                if (wall_clock64() - t0 > AQ_TIMEOUT) { if (err_flag) atomicOr(err_flag, 4); atomicExch(&q[AQ_DONE], 1); break; }
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    return __builtin_amdgcn_readfirstlane(payload);
}
// producer side, any single lane.  The caller has won the tile's membership flag.
__device__ __forceinline__ void aq_push(unsigned long long* __restrict__ ring, int32_t* __restrict__ q, int cap, int payload,
                                        int32_t* __restrict__ err_flag) {
    atomicAdd(&q[AQ_PENDING], 1);
    const unsigned t = (unsigned)atomicAdd(&q[AQ_TAIL], 1);
    unsigned long long* slot = ring + t % (unsigned)cap;
    const unsigned long long e = ((unsigned long long)(t + 1u) << 32) | (unsigned)payload;
    const long long t0 = wall_clock64();
    while (atomicCAS(slot, 0ull, e) != 0ull) {           // the previous user of the slot (ticket t - cap) has not read it yet
        if (wall_clock64() - t0 > AQ_TIMEOUT) { if (err_flag) atomicOr(err_flag, 4); atomicExch(&q[AQ_DONE], 1); break; }
        __builtin_amdgcn_s_sleep(2);
    }
}
__device__ __forceinline__ void aq_finish(int32_t* __restrict__ q) {
    if (atomicSub(&q[AQ_PENDING], 1) == 1) atomicExch(&q[AQ_DONE], 1);
}

// queue <- the work list a launch-driven kernel (or the active scan) has just produced: list[0 .. *count), flags already set
__global__ void __launch_bounds__(256) k_aq_init(const int32_t* __restrict__ count, const int32_t* __restrict__ list,
                                                 unsigned long long* __restrict__ ring, int32_t* __restrict__ q, int cap, int budget_per_item) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(*count, cap);
    if (i < cap) ring[i] = i < n ? (((unsigned long long)(unsigned)(i + 1)) << 32) | (unsigned)list[i] : 0ull;
    if (i == 0) {
        q[AQ_HEAD] = 0; q[AQ_TAIL] = n; q[AQ_PENDING] = n; q[AQ_DONE] = n == 0 ? 1 : 0; q[AQ_VISITS] = 0;
        q[AQ_BUDGET] = (int)min((long long)n * budget_per_item + 4096, (long long)0x3fffffff);
    }
}

// ---- global relabel, asynchronous ---------------------------------------------------------------------------------
struct RelaxWaveLds { int d[RT + 2][RT + 2]; uint32_t m[RT][RT / 4]; };

__global__ void __launch_bounds__(256) k_mf_relax_async(GcDims d, MfTiles tl, uint8_t* __restrict__ rmask, int32_t* __restrict__ dirty,
                                                        const int32_t* __restrict__ rc, int32_t* __restrict__ dist, int32_t* __restrict__ flag,
                                                        unsigned long long* __restrict__ ring, int32_t* __restrict__ q, int cap,
                                                        int32_t* __restrict__ err_flag) {
    constexpr int T = RT, HALO_IT = (RT_HALO + 63) / 64;
    __shared__ RelaxWaveLds lds[4];
    RelaxWaveLds& S = lds[threadIdx.x >> 6];
    const int tiles_per_image = tl.rt_x * tl.rt_y;
    int* sd = &S.d[0][0];
    uint8_t* sm = reinterpret_cast<uint8_t*>(&S.m[0][0]);
    int tile = -1;                                                         // >= 0: the neighbour this wave follows into
    for (;;) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));                                     // keeps the lane arithmetic inside the loop (no hoist + spill)
        if (tile < 0) {
            tile = aq_pop(ring, q, cap, lane, err_flag);
            if (tile < 0) break;
            if (lane == 0) atomicExch(&flag[tile], 0);                     // consumed: a halo change from now on re-queues the tile
            drain();                                                       // ... and the loads below come after it
        }
        const int b = tile / tiles_per_image, tr = tile % tiles_per_image;
        const int tyi = tr / tl.rt_x, txi = tr % tl.rt_x;
        const int tx0 = txi * T, ty0 = tyi * T;
        const size_t base = (size_t)b * d.P;
        const int lx = lane & 31, h = lane >> 5;
        int hv[HALO_IT];
#pragma unroll
        for (int k = 0; k < HALO_IT; ++k) {                                // unconditional loads from clamped addresses
            const int i = min(lane + k * 64, RT_HALO - 1);
            const int gy = ty0 + i / (T + 2) - 1, gx = tx0 + i % (T + 2) - 1;
            hv[k] = ldg(dist + base + (size_t)min(max(gy, 0), d.H - 1) * d.W + min(max(gx, 0), d.W - 1));
        }
        uint32_t mv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)                                       // arc masks do not change during a relabel: plain loads
            mv[r] = rmask[base + (size_t)min(ty0 + 16 * h + r, d.H - 1) * d.W + min(tx0 + lx, d.W - 1)];
        const bool dirty_t = mf_tile_dirty(dirty, tl, b, tyi, txi, lane);  // wave-uniform (ggc_mf_sweep.h)
        MfBorderArcs ba;                                                   // (capacities do not change during a relabel either)
        if (dirty_t) ba.load(d, rc, (size_t)d.B * d.P, base, ty0, tx0, lx, h);
        uint32_t inv_v[4] = {0u, 0u, 0u, 0u}, inv_h[4];                    // bit set = no arc; outside the image: all blocked
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            uint32_t m = ~mv[r] & 0xffu;
            if (dirty_t) m = ba.row(m, r);
            sm[(16 * h + r) * T + lx] = (uint8_t)((tx0 + lx < d.W && ty0 + 16 * h + r < d.H) ? m : 0xffu);
        }
#pragma unroll
        for (int k = 0; k < HALO_IT; ++k) {
            const int i = lane + k * 64;
            const int gy = ty0 + i / (T + 2) - 1, gx = tx0 + i % (T + 2) - 1;
            if (i < RT_HALO) sd[i] = (gx >= 0 && gx < d.W && gy >= 0 && gy < d.H) ? hv[k] : DINF;
        }
        mf_wave_sync();
        if (dirty_t) {      // (a second wave relaxing the same tile meanwhile writes the same bytes)
            if (ty0 + lx < d.H && tx0 + (h ? 31 : 0) < d.W) sm[lx * T + (h ? 31 : 0)] = (uint8_t)ba.col(sm[lx * T + (h ? 31 : 0)], h);
            mf_wave_sync();
            mf_tile_repair(d, tl, rmask, dirty, S, base, b, tyi, txi, ty0, tx0, lx, h, lane);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) inv_v[r >> 2] |= (uint32_t)sm[(16 * h + r) * T + lx] << (8 * (r & 3));
#pragma unroll
        for (int k = 0; k < 4; ++k) inv_h[k] = S.m[lx][4 * h + k];         // H sweep: row lx, columns 16h .. 16h+15
        int old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = S.d[16 * h + r + 1][lx + 1];
        bool settled = false;
        for (int it = 0; it < 4 * T; ++it) {                               // a sweep that changes nothing: fixpoint (a visit capped at 2-6 sweeps and re-queued publishes its border earlier, but costs more visits: 56.4 -> 60.3 / 58.1 / 56.6 ms)
            const int ch = (it & 1) ? relax_sweep_h(S, inv_h, lx, h) : relax_sweep_v(S, inv_v, lx, h);
            mf_wave_sync();
            if (!__any(ch)) { settled = true; break; }
        }
        int nbm = settled ? 0 : 1 << 4;                                    // bit (dy + 1) * 3 + (dx + 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ly = 16 * h + r;
            const int v = S.d[ly + 1][lx + 1];
            if (v != old[r]) {                                             // (pixels outside the image never change: all arcs blocked)
                atomicMin(&dist[base + (size_t)(ty0 + ly) * d.W + tx0 + lx], v);
                const int Lf = lx == 0, Rt = lx == T - 1, U = ly == 0, D = ly == T - 1;
                nbm |= (U & Lf) | U << 1 | (U & Rt) << 2 | Lf << 3 | Rt << 5 | (D & Lf) << 6 | D << 7 | (D & Rt) << 8;
            }
        }
        nbm = mf_wave_or(nbm);
        drain();                                                           // the new labels are at memory before a neighbour is told
        // Neighbours whose halo changed and that are not queued: the first one this wave visits next itself (its flag stays 0,
        // as if popped; a second wave relaxing the same tile meanwhile is harmless — labels only fall, atomically), the
        // others go through the queue.
        int nb = -1;
        if (lane < 9 && (nbm >> lane) & 1) {
            const int ty = tyi + lane / 3 - 1, tx = txi + lane % 3 - 1;
            if (ty >= 0 && ty < tl.rt_y && tx >= 0 && tx < tl.rt_x) nb = b * tiles_per_image + ty * tl.rt_x + tx;
        }
        const bool cand = nb >= 0 && ldg(&flag[nb]) == 0;
        const unsigned long long cm = __ballot(cand);
        int next = -1;
        if (cm) {
            const int fl = __ffsll((long long)cm) - 1;
            next = __shfl(nb, fl, 64);
            if (cand && lane != fl && atomicExch(&flag[nb], 1) == 0) aq_push(ring, q, cap, nb, err_flag);
            drain();                                                       // the entries count as pending before this visit ends
        }
        if (next < 0 && lane == 0) aq_finish(q);                           // (following keeps this visit's pending unit)
        tile = next;
        mf_wave_sync();
    }
}

// ---- push-relabel sweeps, asynchronous --------------------------------------------------------------------------------
constexpr int ST_Q = 1, ST_BUSY = 2;                  // tile state word: queued (or owed a visit) | a wave holds it

// One wave, one 32 x TH tile (TH/2 pixels per lane for loading and write-back).  The tile's active pixels (excess that can
// still reach the sink) are a bitmask, one word per row; a sweep compacts it into a list and hands one pixel to each lane,
// so a sweep costs what the active pixels cost and a visit can afford enough of them to carry excess across the tile.
template <int TH>
struct PushTileLds {
    int ex[32 * TH]; int sk[32 * TH]; int d[TH + 2][34]; int rc[8][32 * TH];
    uint32_t mask[32]; unsigned short list[32 * TH];
    unsigned char streak[32 * TH];      // relabels in a row without a push (a pixel is handled by one lane at a time)
};

// Returns the 9-bit mask of tiles owed a visit: bit (dy + 1) * 3 + (dx + 1) for a neighbour that received excess, bit 4
// when this tile still holds active pixels.
template <int TH>
__device__ int push_tile_visit(const GcDims& d, int tyi, int txi, int inner, int chase, int park, size_t base, size_t BP,
                               int32_t* __restrict__ rc, int32_t* __restrict__ ex, int32_t* __restrict__ snk,
                               int32_t* __restrict__ dist, uint8_t* __restrict__ rmask, PushTileLds<TH>& S, int lane,
                               bool prof, long long (&pv)[4]) {
    constexpr int NPX = TH / 2, HALO = (TH + 2) * 34, HALO_IT = (HALO + 63) / 64;
    const long long tv0 = prof ? wall_clock64() : 0;
    const int lx = lane & 31, r0 = lane >> 5;
    const int x = txi * 32 + lx;
    int e0[NPX], sk0[NPX], r0v[NPX][8];
    // every load of the visit is issued unconditionally from a clamped address, then masked
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int y = tyi * TH + r0 + 2 * j;
        const int pc = min(y, d.H - 1) * d.W + min(x, d.W - 1);
        e0[j] = ldg(ex + base + pc);
        sk0[j] = ldg(snk + base + pc);
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) r0v[j][dir] = ldg(rc + rc_idx(dir, base + pc));
    }
    int* sd = &S.d[0][0];
    int hv[HALO_IT];
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
        const int i = min(lane + k * 64, HALO - 1);
        const int gy = tyi * TH + i / 34 - 1, gx = txi * 32 + i % 34 - 1;
        hv[k] = ldg(dist + base + (size_t)min(max(gy, 0), d.H - 1) * d.W + min(max(gx, 0), d.W - 1));
    }
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
        const int i = lane + k * 64;
        const int gy = tyi * TH + i / 34 - 1, gx = txi * 32 + i % 34 - 1;
        if (i < HALO) sd[i] = (gx >= 0 && gx < d.W && gy >= 0 && gy < d.H) ? hv[k] : DINF;
    }
    if (lane < 32) S.mask[lane] = 0u;
    for (int i = lane; i < 32 * TH / 4; i += 64) reinterpret_cast<uint32_t*>(S.streak)[i] = 0u;
    mf_wave_sync();
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int slot = lane + 64 * j, ly = r0 + 2 * j;
        const bool inb = x < d.W && tyi * TH + ly < d.H;
        if (!inb) { e0[j] = 0; sk0[j] = 0; }
        S.ex[slot] = e0[j];
        S.sk[slot] = sk0[j];
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) {
            if (!inb) r0v[j][dir] = 0;
            S.rc[dir][slot] = r0v[j][dir];
        }
        const bool a = e0[j] > 0 && S.d[ly + 1][lx + 1] < d.P;
        const unsigned long long m = __ballot(a);                          // lanes 0-31: row 2j, lanes 32-63: row 2j + 1
        if (lane == 0) { S.mask[2 * j] = (uint32_t)m; S.mask[2 * j + 1] = (uint32_t)(m >> 32); }
    }
    mf_wave_sync();
    int nbm = 0;
    const long long tv1 = prof ? wall_clock64() : 0;
    int n_sw = 0;
    for (int it = 0; it < inner; ++it) {
        // ---- bitmask -> list (row-major), mask cleared
        // every lane reads all TH row words (broadcast reads) and forms the row offsets itself: no cross-lane step (a prefix
        // scan by __shfl_up is five dependent LDS-crossbar round trips, a third of a sparse sweep)
        uint32_t w = 0u;
        int off = 0, n_act = 0;
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            const uint32_t wr = S.mask[r];
            const int cr = __popc(wr);
            w = r == lane ? wr : w;
            off += r < lane ? cr : 0;
            n_act += cr;
        }
        if (n_act == 0) break;
        ++n_sw;
        mf_wave_sync();                                                    // every lane has read the words before they are cleared
        if (lane < TH) S.mask[lane] = 0u;
        while (w) { const int bit = __ffs(w) - 1; S.list[off++] = (unsigned short)(lane * 32 + bit); w &= w - 1; }
        mf_wave_sync();
        // ---- one active pixel per lane, and the lane CHASES the excess: after a push inside the tile it goes on with the
        // receiving pixel at once (and after a relabel with the same pixel), up to `chase` steps.  A step is one LDS round trip
        // (~0.2 us) where a sweep is six (0.94 us), and in the sparse rounds a visit is mostly one or two units of excess
        // walking across the tile.  Exclusivity: a lane TAKES a pixel's excess with an exchange (two lanes that meet on a pixel:
        // one gets it all, the other gets zero and drops out) and gives back what it could not push.
        for (int k0 = 0; k0 < n_act; k0 += 64) {
            int slot = k0 + lane < n_act ? (int)S.list[k0 + lane] : -1;
            for (int step = 0; step < chase; ++step) {
                if (!__any(slot >= 0)) break;
                if (slot >= 0) {
                    const int ly = slot >> 5, plx = slot & 31;
                    const int e = atomicExch(&S.ex[slot], 0);
                    const int dp = S.d[ly + 1][plx + 1];
                    const int sk = S.sk[slot];
                    int r[8], hq[8];
#pragma unroll
                    for (int dir = 0; dir < 8; ++dir) {
                        r[dir] = __hip_atomic_load(&S.rc[dir][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        hq[dir] = S.d[ly + 1 + dir_dy(dir)][plx + 1 + dir_dx(dir)];
                    }
                    if (e <= 0) {
                        slot = -1;                                     // taken by a lane that walked in, or nothing left
                    } else if (dp >= d.P) {
                        atomicAdd(&S.ex[slot], e); slot = -1;          // cannot reach the sink: the excess stays where it is
                    } else {
                        int hmin = sk > 0 ? 0 : DINF, best = sk > 0 ? 8 : -1, rb = 0;
#pragma unroll
                        for (int dir = 0; dir < 8; ++dir) {
                            const bool ok = r[dir] > 0 && hq[dir] < hmin;
                            hmin = ok ? hq[dir] : hmin; best = ok ? dir : best; rb = ok ? r[dir] : rb;
                        }
                        if (best >= 0 && dp > hmin) {
                            S.streak[slot] = 0;
                            if (best == 8) {
                                const int dl = min(e, sk);
                                S.sk[slot] = sk - dl;                  // only the holder of the pixel's excess touches its sink link
                                if (e - dl > 0) atomicAdd(&S.ex[slot], e - dl); else slot = -1;   // sink link full: on to the neighbours
                            } else {
                                const int dl = min(e, rb), rem = e - dl;
                                atomicSub(&S.rc[best][slot], dl);
                                if (rem > 0) { atomicAdd(&S.ex[slot], rem); atomicOr(&S.mask[ly], 1u << plx); }   // the rest: next sweep
                                const int qlx = plx + dir_dx(best), qly = ly + dir_dy(best);
                                if (qlx >= 0 && qlx < 32 && qly >= 0 && qly < TH) {
                                    const int qt = qly * 32 + qlx;
                                    atomicAdd(&S.rc[best ^ 1][qt], dl);
                                    atomicAdd(&S.ex[qt], dl);
                                    slot = qt;                          // follow the flow
                                } else {                                // across the tile edge: straight to memory
                                    const int gy = tyi * TH + qly, gx = txi * 32 + qlx;
                                    const size_t qg = base + (size_t)gy * d.W + gx;
                                    atomicAdd(&rc[rc_idx((best ^ 1), qg)], dl);
                                    atomicAdd(&ex[qg], dl);
                                    const int tdy = qly < 0 ? -1 : (qly >= TH ? 1 : 0), tdx = qlx < 0 ? -1 : (qlx >= 32 ? 1 : 0);
                                    nbm |= 1 << ((tdy + 1) * 3 + tdx + 1);
                                    slot = rem > 0 ? slot : -1;
                                }
                            }
                        } else {
                            const int nd = (best >= 0 && hmin < DINF) ? hmin + 1 : DINF;
                            S.d[ly + 1][plx + 1] = nd;
                            atomicAdd(&S.ex[slot], e);                 // give it back; with the new label the next step can push
                            // A pixel that only climbs — `park` relabels in a row without a push — is left alone for the rest of
                            // this visit and does not keep the tile on the queue; the next global relabel gives it an exact label
                            // (or none) and the active scan finds it again.  Off by default (GGC_MF_ASYNC_PARK, 255): measured
                            // without effect at 2-6 — trapped excess does not climb in place, it is pushed back and forth between
                            // the pixels of its pocket, every push resetting the streak.
                            const int sr = S.streak[slot] + 1;
                            S.streak[slot] = (unsigned char)min(sr, 255);
                            if (nd >= d.P || sr >= park) slot = -1;
                        }
                    }
                }
            }
            if (slot >= 0) atomicOr(&S.mask[slot >> 5], 1u << (slot & 31));     // step limit: the pixel is looked at again next sweep
            mf_wave_sync();
        }
    }
    const long long tv2 = prof ? wall_clock64() : 0;
    if (prof) { pv[0] += tv1 - tv0; pv[1] += tv2 - tv1; pv[2] += n_sw; }
    int left = 0;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int slot = lane + 64 * j, ly = r0 + 2 * j;
        const int y = tyi * TH + ly;
        const bool inb = x < d.W && y < d.H;
        const size_t p = base + (size_t)min(y, d.H - 1) * d.W + min(x, d.W - 1);
        const int e1 = S.ex[slot], sk1 = S.sk[slot], d1 = S.d[ly + 1][lx + 1];
        int r1[8];
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) r1[dir] = S.rc[dir][slot];
        if (!inb) continue;
        // everything returns as an atomic read-modify-write: deltas where a neighbouring tile's wave may have added meanwhile
        // (excess, reverse arcs), exchanges for what only the tile's holder writes (sink link, label)
        if (e1 != e0[j]) atomicAdd(&ex[p], e1 - e0[j]);
        int m1 = 0, chg = 0;
#pragma unroll
        for (int dir = 0; dir < 8; ++dir) {
            m1 |= (r1[dir] > 0) ? (1 << dir) : 0;
            if (r1[dir] != r0v[j][dir]) { chg = 1; atomicAdd(&rc[rc_idx(dir, p)], r1[dir] - r0v[j][dir]); }
        }
        if (chg) {
            // read by the next relabel only, but a tile can be visited from several XCDs within this launch: two plain stores to
            // one byte would sit in two L2s and reach memory in either order.  Atomics on the byte's word (clear, then set).
            uint32_t* w = reinterpret_cast<uint32_t*>(rmask + (p & ~(size_t)3));
            const int sh = 8 * (int)(p & 3);
            atomicAnd(w, ~(0xffu << sh));
            atomicOr(w, (uint32_t)m1 << sh);
        }
        if (sk1 != sk0[j]) atomicExch(&snk[p], sk1);
        // the label: compare with the halo copy's origin is not kept, so write when the pixel was relabelled (d only rises here)
        left |= (e1 > 0 && d1 < d.P && S.streak[slot] < park) ? 1 : 0;
    }
    // labels: a pixel's label is written when it differs from what memory held at load time
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
        const int i = lane + k * 64;
        const int hy = i / 34, hx = i % 34;
        if (i < HALO && hy >= 1 && hy <= TH && hx >= 1 && hx <= 32) {
            const int gy = tyi * TH + hy - 1, gx = txi * 32 + hx - 1;
            if (gx < d.W && gy < d.H && sd[i] != hv[k]) atomicExch(&dist[base + (size_t)gy * d.W + gx], sd[i]);
        }
    }
    if (__any(left)) nbm |= 1 << 4;
    return mf_wave_or(nbm);
}

template <int TH>
__global__ void __launch_bounds__(64) k_mf_push_async(GcDims d, int bt_x, int bt_y, int pt_y, int inner, int chase, int park, int gen_max, int follow, int32_t* __restrict__ dirty, int32_t* __restrict__ rc,
                                                      int32_t* __restrict__ ex, int32_t* __restrict__ snk, int32_t* __restrict__ dist,
                                                      uint8_t* __restrict__ rmask, int32_t* __restrict__ st, unsigned long long* __restrict__ ring,
                                                      int32_t* __restrict__ q, int cap, int32_t* __restrict__ err_flag,
                                                      long long* __restrict__ prof) {
    __shared__ PushTileLds<TH> S;
    const int tiles_per_image = bt_x * bt_y;
    const size_t BP = (size_t)d.B * d.P;
    const int budget = ldg(&q[AQ_BUDGET]);
    int tile = -1, gen = 0;
    // GGC_MF_TRACE (prof != null): where a wave's time goes — waiting for a queue entry, the visit, the hand-over — in
    // wall_clock64 ticks, kept in registers and added up once when the wave leaves
    long long p_wait = 0, p_visit = 0, p_hand = 0, p_n = 0, p_follow = 0, pv[4] = {0, 0, 0, 0};
    const long long t_start = prof ? wall_clock64() : 0;
    for (;;) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const long long t_0 = prof ? wall_clock64() : 0;
        if (tile < 0) {
            const int payload = aq_pop(ring, q, cap, lane, err_flag);
            if (payload < 0) { if (prof) p_wait += wall_clock64() - t_0; break; }
            tile = payload & 0xffffff; gen = payload >> 24;
        } else if (prof) ++p_follow;
        // A tile that is queued (or followed into) is never busy: it is only queued from the idle state, and its holder
        // re-queues it after letting go.  Busy from here on; excess that arrives meanwhile sets `queued` again.
        int over = 0;
        if (lane == 0) {
            atomicExch(&st[tile], ST_BUSY);
            over = atomicAdd(&q[AQ_VISITS], 1) >= budget;                  // runaway guard: the rest waits for the next relabel
            if (over) atomicExch(&q[AQ_DONE], 1);
        }
        drain();                                                           // ... before the tile is loaded
        const long long t_1 = prof ? wall_clock64() : 0;
        const int b = tile / tiles_per_image, tr = tile % tiles_per_image;
        const int tyi = tr / bt_x, txi = tr % bt_x;
        const int nbm = push_tile_visit<TH>(d, tyi, txi, inner, chase, park, (size_t)b * d.P, BP, rc, ex, snk, dist, rmask, S, lane, prof != nullptr, pv);
        drain();                                                           // the write-back is at memory
        const long long t_2 = prof ? wall_clock64() : 0;
        const bool left = (nbm >> 4) & 1;
        int nb = -1;
        bool cand = false;
        if (lane == 4) {
            nb = tile;
            const int old = atomicExch(&st[tile], left ? ST_Q : 0);        // let go; still active: owed a visit, ours to arrange
            if (left) cand = true;
            else if (old & ST_Q) cand = (atomicOr(&st[tile], ST_Q) & (ST_Q | ST_BUSY)) == 0;   // excess arrived while we held it
        } else if (lane < 9 && (nbm >> lane) & 1) {
            const int ty = tyi + lane / 3 - 1, tx = txi + lane % 3 - 1;
            if (ty >= 0 && ty < bt_y && tx >= 0 && tx < bt_x) {
                nb = b * tiles_per_image + ty * bt_x + tx;
                // its border arcs may have been re-opened: the 32x8 push tiles it consists of are dirty for the next relabel
                for (int k = 0; k < TH / PT_H; ++k)
                    if (ty * (TH / PT_H) + k < pt_y) atomicOr(&dirty[(size_t)b * bt_x * pt_y + (size_t)(ty * (TH / PT_H) + k) * bt_x + tx], 1);
                cand = (atomicOr(&st[nb], ST_Q) & (ST_Q | ST_BUSY)) == 0;  // idle: ours to arrange; busy: its holder re-queues it
            }
        }
        if (gen + 1 >= gen_max) cand = false;                              // chain length reached: left for the next relabel
        const unsigned long long cm = __ballot(cand);
        int next = -1;
        if (cm) {
            const int fl = __ffsll((long long)cm) - 1;
            if (follow) next = __shfl(nb, fl, 64);
            if (cand && (lane != fl || !follow)) aq_push(ring, q, cap, ((gen + 1) << 24) | nb, err_flag);
            drain();
        }
        if (next < 0 && lane == 0) aq_finish(q);
        tile = next; gen = gen + 1;
        mf_wave_sync();
        if (prof) { p_wait += t_1 - t_0; p_visit += t_2 - t_1; p_hand += wall_clock64() - t_2; ++p_n; }
    }
    if (prof && (threadIdx.x & 63) == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(prof) + (blockIdx.x & 63) * 8;
        atomicAdd(&o[0], (unsigned long long)p_wait); atomicAdd(&o[1], (unsigned long long)p_visit); atomicAdd(&o[2], (unsigned long long)p_hand);
        atomicAdd(&o[3], (unsigned long long)p_n); atomicAdd(&o[4], (unsigned long long)p_follow);
        atomicAdd(&o[5], (unsigned long long)(wall_clock64() - t_start)); atomicAdd(&o[6], 1ull);
        unsigned long long* o2 = o + 64 * 8;                               // second table: inside the visit
        atomicAdd(&o2[0], (unsigned long long)pv[0]); atomicAdd(&o2[1], (unsigned long long)pv[1]); atomicAdd(&o2[2], (unsigned long long)pv[2]);
    }
}

// queue <- the big tiles that hold the 32x8 push tiles of list[0 .. *count) (the active scan's list); ring, q and st zeroed before
__global__ void __launch_bounds__(256) k_aq_fill_big(const int32_t* __restrict__ count, const int32_t* __restrict__ list, int pt_x, int pt_y,
                                                     int th, int bt_x, int bt_y, int32_t* __restrict__ st,
                                                     unsigned long long* __restrict__ ring, int32_t* __restrict__ q, int budget_per_item) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *count;
    if (i == 0) q[AQ_BUDGET] = (int)min((long long)n * budget_per_item + 4096, (long long)0x3fffffff);
    if (i >= n) return;
    const int t = list[i], b = t / (pt_x * pt_y), tr = t % (pt_x * pt_y);
    const int big = b * bt_x * bt_y + ((tr / pt_x) * PT_H / th) * bt_x + tr % pt_x;
    if (atomicOr(&st[big], ST_Q) == 0) {
        const int slot = atomicAdd(&q[AQ_TAIL], 1);
        atomicAdd(&q[AQ_PENDING], 1);
        ring[slot] = (((unsigned long long)(unsigned)(slot + 1)) << 32) | (unsigned)big;
    }
}

} // namespace

int maxflow_relax_async(ggc_ctx* ctx, hipStream_t st, const GcDims& d, const MfTiles& tl, uint8_t* rmask, int32_t* dirty, const int32_t* rc, int32_t* dist,
                        const int32_t* count, const int32_t* list, int32_t* flag, unsigned long long* ring, int32_t* q, int cap,
                        int grid, int32_t* err_flag) {
    hipLaunchKernelGGL(k_aq_init, dim3(cdiv(cap, 256)), dim3(256), 0, st, count, list, ring, q, cap, 0);
    hipLaunchKernelGGL(k_mf_relax_async, dim3(grid), dim3(256), 0, st, d, tl, rmask, dirty, rc, dist, flag, ring, q, cap, err_flag);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

int maxflow_push_async(ggc_ctx* ctx, hipStream_t st, const GcDims& d, const MfTiles& tl, int th, int inner, int gen_max, int32_t* rc,
                       int32_t* ex, int32_t* snk, int32_t* dist, uint8_t* rmask, int32_t* dirty, const int32_t* count, const int32_t* list, int n_list_max, int32_t* state,
                       unsigned long long* ring, int32_t* q, int waves, int32_t* err_flag, long long* prof) {
    th = th >= 32 ? 32 : (th >= 16 ? 16 : 8);
    const int bt_x = tl.pt_x, bt_y = cdiv(d.H, th), cap = bt_x * bt_y * d.B;
    gen_max = std::min(gen_max, 127);
    // Fixed after the round-2 measurements: the wave that finishes a visit FOLLOWS the front into one of the
    // tiles it would have queued (74.6 -> 69.9 ms per stage); chasing a unit of excess inside a sweep (chase > 1) and
    // parking pixels that only climb (park < 255) measured slower / without effect, so both stay off.
    const int follow = 1, park = 255, chase = 1;
    mf_zero3(st, reinterpret_cast<int32_t*>(ring), (size_t)cap * 2, q, AQ_WORDS, state, (size_t)cap);
    hipLaunchKernelGGL(k_aq_fill_big, dim3(cdiv(n_list_max, 256)), dim3(256), 0, st, count, list, tl.pt_x, tl.pt_y, th, bt_x, bt_y, state, ring, q,
                       gen_max);
    if (th == 32)
        hipLaunchKernelGGL(k_mf_push_async<32>, dim3(waves), dim3(64), 0, st, d, bt_x, bt_y, tl.pt_y, inner, chase, park, gen_max, follow, dirty, rc, ex, snk, dist, rmask, state, ring, q, cap,
                           err_flag, prof);
    else if (th == 8)
        hipLaunchKernelGGL(k_mf_push_async<8>, dim3(waves), dim3(64), 0, st, d, bt_x, bt_y, tl.pt_y, inner, chase, park, gen_max, follow, dirty, rc, ex, snk, dist, rmask, state, ring, q, cap,
                           err_flag, prof);
    else
        hipLaunchKernelGGL(k_mf_push_async<16>, dim3(waves), dim3(64), 0, st, d, bt_x, bt_y, tl.pt_y, inner, chase, park, gen_max, follow, dirty, rc, ex, snk, dist, rmask, state, ring, q, cap,
                           err_flag, prof);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

} // namespace ggc
