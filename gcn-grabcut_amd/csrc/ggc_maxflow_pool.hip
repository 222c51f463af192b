// ggc_maxflow_pool.hip — the max-flow of ggc_maxflow.hip as ONE launch per GrabCut iteration: a pool of resident waves
// works through per-image task lists, and every decision the host used to take (is the relabel front empty?  how many
// active pixels are left?  another push pass, another relabel, done?) is taken on the device by whichever wave finishes
// the last task of an image's pass.  Same algorithm and tiles as ggc_maxflow.hip, same canonical result.
//
// Why not a workgroup per image (ggc_maxflow_image.hip): images differ 4x in work, so a launch waited for its slowest
// image with 255 CUs idle.  Why not a grid barrier per pass: it needs the whole grid co-resident and serialises images.
//
//  * an image is a state machine  INIT -> RELAX* -> SCAN -> (PUSH* -> INIT ...) | done.  A pass is a set of independent
//    tile tasks; {epoch, next index} is one 64-bit word grabbed with an atomic add, {epoch, phase, n} another.  The
//    wave whose task completes the pass (device-scope counter) compacts the image's next-tile bitmap into its list,
//    picks the next phase and publishes the new epoch.  NO wave ever waits for another one: a wave without a task
//    polls, and every poll it makes can only be satisfied by waves that are already running, so the launch completes
//    with any number of resident workgroups (all spins are bounded and report through the error word).
//  * who may touch an image: per-XCD L2s are not coherent with each other, so an image is CLAIMED (atomic compare and
//    swap) by the XCD of the first wave that needs work, and from then on only waves that read the same HW_REG_XCC_ID
//    touch it.  Inside one XCD plain stores (write-through L1), drained with s_waitcnt vmcnt(0) before the completion
//    counter is bumped, are visible to sc1 (L2-served) loads of every CU: tools/micro/xcd_visibility.hip measures 0
//    stale values of 5e8 for that pattern on MI355X (and 100 % stale across XCDs, which is why the claim exists).
//    Every load of image data in here is an sc1 load; the claim, the counters and the bitmaps are device-scope atomics.
//  * a WAVE is the worker: it stages a tile in its private LDS slice (relabel: 32x32 labels + halo + arc masks; push:
//    32x8 excess / sink links / 8 residual planes / labels + halo), works wave-synchronously and writes back.  There is
//    no workgroup barrier in the kernel.
#include "ggc_gc.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace ggc {
namespace {

constexpr int RT = MF_RT, PT_W = MF_PT_W, PT_H = MF_PT_H, PT_N = PT_W * PT_H;
constexpr int MAX_TILES = 8192, BM_WORDS = MAX_TILES / 32;
constexpr int PT_PX = PT_N / 64;                     // pixels per lane in a push tile (4)
constexpr int RT_HALO = (RT + 2) * (RT + 2), PT_HALO = (PT_H + 2) * (PT_W + 2);
constexpr int WG_WAVES = 4, WG_T = WG_WAVES * 64, WG_PER_CU = 3;

enum Phase : int { PH_INIT = 0, PH_RELAX = 1, PH_SCAN = 2, PH_PUSH = 3 };
constexpr int OWNER_FREE = -1, OWNER_DONE = 64;

struct Sched { int passes0, passes, inner, tail_active, tail_passes, tail_inner, max_rounds, max_idle; };

// per image; head / nlist / owner / done are only ever touched with atomics
struct alignas(128) PoolCtl {
    unsigned long long head;        // {epoch : 32 | next task index : 32}
    unsigned long long nlist;       // {epoch : 32 | phase : 4 | sweeps per visit : 8 | tasks : 20}
    int owner, done, active, round, passes_left, pad[23];
};
struct PoolGlobal { int n_open, finished, error, pad; int xcd_count[16]; };

struct PushLds { int ex[PT_N]; int sk[PT_N]; int d[PT_H + 2][PT_W + 2]; int rc[8][PT_N]; unsigned short act[PT_N]; };
struct RelaxLds { int d[RT + 2][RT + 2]; uint32_t m[RT][RT / 4]; int o[RT][RT]; };
union WaveLds { PushLds push; RelaxLds relax; };

__device__ __forceinline__ int ldg(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ldg(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ldg8(const uint8_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ldg64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// control words are only ever WRITTEN by atomic read-modify-writes: those land in memory and drop the L2 line, so that an sc1
// load of any XCD returns them (tools/micro/xcd_atomics.hip: 0 stale of 1.6e8, same XCD and across XCDs)
__device__ __forceinline__ void stg(int32_t* p, int v) { atomicExch(p, v); }
__device__ __forceinline__ void stg64(unsigned long long* p, unsigned long long v) { atomicExch(p, v); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // this wave's stores and atomics are in L2
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ void wave_sync() {        // LDS traffic of one wave is in order: only the compiler needs telling
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long bcast64(unsigned long long v) {      // lane 0's value, as a wave-uniform
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void flag_tile(uint32_t* __restrict__ bm, int tile) { atomicOr(&bm[tile >> 5], 1u << (tile & 31)); }   // the image's NEXT list

// ---- relabel tile visit -----------------------------------------------------------------------------------------
// d(p) = 1 + min over residual arcs p -> q of d(q), relaxed to the tile's fixpoint against a fixed halo.  A sweep where
// every pixel looks at its 8 neighbours once moves the BFS front one pixel (32+ sweeps per tile, each a chain of LDS round
// trips).  Here a lane owns 16 consecutive pixels of one column (V sweep) or of one row (H sweep): it reads its 18x3
// window in one batch, runs a forward and a backward pass over its pixels IN REGISTERS (a front travels the whole
// segment in one pass), and stores what changed.  Alternating V and H sweeps carry a front across the tile in a
// handful of sweeps; the arithmetic is branch-free (a missing arc ORs the "infinite" bit into the neighbour's label).
__device__ __forceinline__ int gated(int v, uint32_t inv, int bit) {        // v if the arc exists, >= DINF otherwise
    return (__builtin_amdgcn_sbfe((int)inv, bit, 1) & DINF) | v;
}
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int relax_px(int c, uint32_t inv, int pos, int lf, int rt, int up, int dn, int ul, int dr, int ur, int dl) {
    const int nd = min3i(min3i(gated(lf, inv, pos), gated(rt, inv, pos + 1), gated(up, inv, pos + 2)),
                         min3i(gated(dn, inv, pos + 3), gated(ul, inv, pos + 4), gated(dr, inv, pos + 5)),
                         min(gated(ur, inv, pos + 6), gated(dl, inv, pos + 7)));
    return min(c, nd + 1);
}

typedef RelaxLds RelaxTile;                          // labels + halo, inverted arc masks (1 byte per pixel)

// lane = (column lx, half h): pixels (rows 16h .. 16h+15, column lx).  Returns 1 when a label changed.
__device__ __forceinline__ int relax_sweep_v(RelaxTile& S, const uint32_t (&inv_in)[4], int lx, int h) {
    // the per-arc gate words are loop invariants of the caller's sweep loop: hide the masks from the optimiser, or it hoists
    // 128 of them out of the loop and spills
    uint32_t inv[4] = {inv_in[0], inv_in[1], inv_in[2], inv_in[3]};
    asm volatile("" : "+v"(inv[0]), "+v"(inv[1]), "+v"(inv[2]), "+v"(inv[3]));
    int w[18][3];
#pragma unroll
    for (int a = 0; a < 18; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) w[a][c] = S.d[16 * h + a][lx + c];
    uint32_t chg = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int a = r + 1;
        const int nv = relax_px(w[a][1], inv[r >> 2], 8 * (r & 3), w[a][0], w[a][2], w[a - 1][1], w[a + 1][1], w[a - 1][0], w[a + 1][2],
                                w[a - 1][2], w[a + 1][0]);
        chg |= (nv != w[a][1]) ? 1u << r : 0u;
        w[a][1] = nv;
    }
#pragma unroll
    for (int r = 15; r >= 0; --r) {
        const int a = r + 1;
        const int nv = relax_px(w[a][1], inv[r >> 2], 8 * (r & 3), w[a][0], w[a][2], w[a - 1][1], w[a + 1][1], w[a - 1][0], w[a + 1][2],
                                w[a - 1][2], w[a + 1][0]);
        chg |= (nv != w[a][1]) ? 1u << r : 0u;
        w[a][1] = nv;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((chg >> r) & 1u) S.d[16 * h + r + 1][lx + 1] = w[r + 1][1];
    return chg != 0u;
}
// lane = (row ly, half h): pixels (row ly, columns 16h .. 16h+15)
__device__ __forceinline__ int relax_sweep_h(RelaxTile& S, const uint32_t (&inv_in)[4], int ly, int h) {
    uint32_t inv[4] = {inv_in[0], inv_in[1], inv_in[2], inv_in[3]};
    asm volatile("" : "+v"(inv[0]), "+v"(inv[1]), "+v"(inv[2]), "+v"(inv[3]));
    int w[3][18];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 18; ++c) w[a][c] = S.d[ly + a][16 * h + c];
    uint32_t chg = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = k + 1;
        const int nv = relax_px(w[1][c], inv[k >> 2], 8 * (k & 3), w[1][c - 1], w[1][c + 1], w[0][c], w[2][c], w[0][c - 1], w[2][c + 1],
                                w[0][c + 1], w[2][c - 1]);
        chg |= (nv != w[1][c]) ? 1u << k : 0u;
        w[1][c] = nv;
    }
#pragma unroll
    for (int k = 15; k >= 0; --k) {
        const int c = k + 1;
        const int nv = relax_px(w[1][c], inv[k >> 2], 8 * (k & 3), w[1][c - 1], w[1][c + 1], w[0][c], w[2][c], w[0][c - 1], w[2][c + 1],
                                w[0][c + 1], w[2][c - 1]);
        chg |= (nv != w[1][c]) ? 1u << k : 0u;
        w[1][c] = nv;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if ((chg >> k) & 1u) S.d[ly + 1][16 * h + k + 1] = w[1][k + 1];
    return chg != 0u;
}

// one visit of a 32x32 relabel tile by one wave: relax to the local fixpoint, write back, flag neighbours whose halo changed
__device__ void relax_visit(const GcDims& d, const MfTiles& tl, int tile, size_t base, size_t BP,
                                                      const int32_t* __restrict__ rc, int32_t* __restrict__ dist,
                                                      const uint8_t* __restrict__ rmask, RelaxTile& S, uint32_t* __restrict__ bm, int lane_in) {
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
        mv[r] = ldg8(rmask + base + (size_t)min(gy, d.H - 1) * d.W + min(gx, d.W - 1));
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
    // a full sweep that changes nothing has checked every pixel against unchanged neighbours: fixpoint
    for (int it = 0; it < 4 * RT; ++it) {
        const int ch = (it & 1) ? relax_sweep_h(S, inv_h, lx, h) : relax_sweep_v(S, inv_v, lx, h);
        wave_sync();
        if (!__any(ch)) { settled = true; break; }
    }
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
}

// ---- push tile visit ----------------------------------------------------------------------------------------------
// One wave, one 32x8 tile (4 pixels per lane for loading and write-back).  A sweep costs the wave its whole instruction
// stream whenever ANY lane has an active pixel, and only ~10 % of a visited tile's pixels are active: so every sweep
// first compacts the active pixels into an LDS list (ballot + mbcnt) and then hands ONE active pixel to each lane.
// The pixel's 8 residual capacities and 8 neighbour labels are read in one batch; the arg-min is branch-free.
__device__ void push_visit(const GcDims& d, const MfTiles& tl, int tile, int inner, size_t base, size_t BP,
                                                     int32_t* __restrict__ rc, int32_t* __restrict__ ex, int32_t* __restrict__ snk,
                                                     int32_t* __restrict__ dist, uint8_t* __restrict__ rmask, PushLds& S, uint32_t* __restrict__ bm,
                                                     int lane_in) {
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
        sk0[j] = ldg(snk + base + pc);
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
}


// ---- small whole-tile tasks ----------------------------------------------------------------------------------------
// INIT task: start of a global relabel for one 32x32 tile: d = 1 next to the sink, infinity elsewhere.  The arc masks
// (rmask bit dir = residual arc p -> nb(dir)) are NOT rebuilt from the 8 capacity planes: k_build_graph writes them and
// every push visit keeps them current for the pixels it owns (relax_visit re-reads the few bits another tile can change).
// A tile without a pixel away from the sink needs no relabel visit.
__device__ void init_task(const GcDims& d, const MfTiles& tl, int tile, size_t base, const int32_t* __restrict__ snk,
                          int32_t* __restrict__ dist, uint32_t* __restrict__ bm, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int ty0 = (tile / tl.rt_x) * RT, tx0 = (tile % tl.rt_x) * RT;
    const int lx = lane & 31, h = lane >> 5;
    const int gx = tx0 + lx;
    int s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = ldg(snk + base + (size_t)min(ty0 + 16 * h + r, d.H - 1) * d.W + min(gx, d.W - 1));
    int far = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gy = ty0 + 16 * h + r;
        if (gx < d.W && gy < d.H) { dist[base + (size_t)gy * d.W + gx] = s[r] > 0 ? 1 : DINF; far |= s[r] <= 0; }
    }
    if (__any(far) && lane == 0) flag_tile(bm, tile);
}

// SCAN task: active pixels (excess that can still reach the sink) of one 32x32 tile: count them, put their push tiles
// (4 bands of 8 rows) on the image's next list
__device__ void scan_task(const GcDims& d, const MfTiles& tl, int tile, size_t base, const int32_t* __restrict__ ex,
                          const int32_t* __restrict__ dist, uint32_t* __restrict__ bm, int* __restrict__ active, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tyi = tile / tl.rt_x, txi = tile % tl.rt_x;
    const int ty0 = tyi * RT, tx0 = txi * RT;
    const int lx = lane & 31, h = lane >> 5;
    const int gx = tx0 + lx;
    int e[16], dd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t i = base + (size_t)min(ty0 + 16 * h + r, d.H - 1) * d.W + min(gx, d.W - 1);
        e[r] = ldg(ex + i); dd[r] = ldg(dist + i);
    }
    int n = 0, band = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool a = gx < d.W && ty0 + 16 * h + r < d.H && e[r] > 0 && dd[r] < DINF;
        n += a ? 1 : 0;
        band |= a ? (1 << (r >> 3)) : 0;
    }
    const unsigned long long b0 = __ballot(band & 1), b1 = __ballot(band & 2);
    // lanes 0..31 hold rows 0..15 (bands 0, 1), lanes 32..63 rows 16..31 (bands 2, 3)
    const int bands = ((unsigned)b0 ? 1 : 0) | ((unsigned)b1 ? 2 : 0) | ((unsigned)(b0 >> 32) ? 4 : 0) | ((unsigned)(b1 >> 32) ? 8 : 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane < 4 && (bands >> lane) & 1) {
        const int py = tyi * (RT / PT_H) + lane;
        if (py < tl.pt_y) flag_tile(bm, py * tl.pt_x + txi);
    }
    if (lane == 0 && n) atomicAdd(active, n);
}

// ---- pass transition ---------------------------------------------------------------------------------------------------
// bitmap -> list (ascending tile order), bitmap cleared; one wave, 4 words per lane
__device__ int compact(uint32_t* __restrict__ bm, int32_t* __restrict__ list, int lane) {
    uint32_t w[4];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { w[k] = atomicExch(bm + lane * 4 + k, 0u); c += __popc(w[k]); }     // read and clear
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    int at = incl - c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t v = w[k];
        while (v) { const int bit = __ffs(v) - 1; list[at++] = (lane * 4 + k) * 32 + bit; v &= v - 1; }
    }
    return __shfl(incl, 63, 64);
}

__device__ __forceinline__ unsigned long long pack_nlist(unsigned epoch, int phase, int inner, int n) {
    return ((unsigned long long)epoch << 32) | ((unsigned long long)(phase & 15) << 28) | ((unsigned long long)(inner & 255) << 20) | (unsigned)(n & 0xfffff);
}

// Called by the wave whose task completed pass `epoch` of image b (every store and atomic of the pass is in L2).
__device__ void finish_pass(const MfTiles& tl, const Sched& sc, PoolCtl* __restrict__ c, PoolGlobal* __restrict__ G, uint32_t* __restrict__ bm,
                            int32_t* __restrict__ list, unsigned epoch, int phase, int my_x, int32_t* __restrict__ err_flag, int lane) {
    const int n_rt = tl.rt_x * tl.rt_y;
    int nphase = phase, n = 0, inner = 0;
    bool finished = false;
    if (phase == PH_INIT || phase == PH_RELAX) {
        n = compact(bm, list, lane);
        if (n > 0) nphase = PH_RELAX; else { nphase = PH_SCAN; n = n_rt; }
    } else if (phase == PH_SCAN) {
        const int active = ldg(&c->active);
        n = compact(bm, list, lane);                       // push tiles that hold an active pixel
        if (active == 0) finished = true;
        else {
            // few active pixels: their labels stay exact, so more (cheap) passes beat another global relabel
            const int round = ldg(&c->round);
            const bool tail = active <= sc.tail_active;
            inner = tail ? sc.tail_inner : sc.inner;
            if (lane == 0) { stg(&c->passes_left, tail ? sc.tail_passes : (round == 0 ? sc.passes0 : sc.passes)); stg(&c->active, 0); }
            nphase = PH_PUSH;
        }
    } else {                                               // PH_PUSH
        n = compact(bm, list, lane);
        const int left = ldg(&c->passes_left) - 1;
        const unsigned long long cur = ldg64(&c->nlist);
        inner = (int)((cur >> 20) & 255);
        if (n > 0 && left > 0) { if (lane == 0) stg(&c->passes_left, left); }
        else {                                             // next round: global relabel (compact() has emptied the bitmap)
            const int round = ldg(&c->round) + 1;
            if (lane == 0) stg(&c->round, round);
            if (round >= sc.max_rounds) { if (lane == 0) atomicOr(err_flag, 1); finished = true; }
            nphase = PH_INIT; n = n_rt;
        }
    }
    if (finished) {
        drain();
        if (lane == 0) { stg(&c->owner, OWNER_DONE); atomicSub(&G->xcd_count[my_x], 1); atomicAdd(&G->finished, 1); }
        return;
    }
    if (lane == 0) stg(&c->done, 0);
    drain();                                               // list, bitmap and control fields first, then the new epoch
    if (lane == 0) { stg64(&c->nlist, pack_nlist(epoch + 1, nphase, inner, n)); drain(); stg64(&c->head, (unsigned long long)(epoch + 1) << 32); }
}

__global__ void k_pool_init(int B, int n_rt, const int32_t* __restrict__ state, PoolCtl* __restrict__ ctl, PoolGlobal* __restrict__ G) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    PoolCtl& c = ctl[b];
    c.head = 1ull << 32;
    c.nlist = pack_nlist(1u, PH_INIT, 0, n_rt);
    c.owner = state[b] ? OWNER_DONE : OWNER_FREE;
    c.done = 0; c.active = 0; c.round = 0; c.passes_left = 0;
    if (!state[b]) atomicAdd(&G->n_open, 1);
}

template <bool PROF>
__global__ void __launch_bounds__(WG_T, WG_PER_CU) k_mf_pool(GcDims d, MfTiles tl, Sched sc, int cap, int32_t* __restrict__ rc,
                                                           int32_t* __restrict__ ex, int32_t* __restrict__ snk, int32_t* __restrict__ dist,
                                                           uint8_t* __restrict__ rmask, PoolCtl* __restrict__ ctl, PoolGlobal* __restrict__ G,
                                                           uint32_t* __restrict__ bms, int32_t* __restrict__ lists, int list_stride,
                                                           int32_t* __restrict__ err_flag, unsigned long long* __restrict__ prof) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // trace build: wall-clock ticks (100 MHz) this wave spent per task kind [0..3], in pass transitions [4], without a task [5];
    // task counts [6..9]
    long long pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = PROF ? wall_clock64() : 0;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds& S = reinterpret_cast<WaveLds*>(smem_raw)[wv];
    const int my_x = xcc_id();
    const size_t BP = (size_t)d.B * d.P;
    const int n_chunks = (d.B + 63) / 64;
    int chunk0 = (blockIdx.x * WG_WAVES + wv) % n_chunks;      // where this wave looks first: it stays with the image that fed it last
    long long idle = 0;
    // One attempt to take a task of image b: true when a task was run (and, if it was the pass' last, the pass advanced).
    auto try_take = [&](int b) -> bool {
        PoolCtl* c = ctl + b;
        unsigned long long h = 0;
        if (lane == 0) h = atomicAdd(&c->head, 1ull);
        drain();                                            // the add has returned BEFORE the pass descriptor is read: an index
        h = bcast64(h);                                     // taken from a newer epoch than the descriptor would be lost
        const unsigned long long nl = bcast64(ldg64(&c->nlist));
        const unsigned epoch = (unsigned)(h >> 32);
        const int i = (int)(unsigned)h, n = (int)(nl & 0xfffff), phase = (int)((nl >> 28) & 15), inner = (int)((nl >> 20) & 255);
        if ((unsigned)(nl >> 32) != epoch || i >= n) return false;             // the pass ran out (or ended) in between
        // ---- task i of pass `epoch`: nothing of this pass can be republished before this wave reports it done
        const size_t base = (size_t)b * d.P;
        uint32_t* bm = bms + (size_t)b * BM_WORDS;
        int32_t* list = lists + (size_t)b * list_stride;
        if (PROF) { const long long t = wall_clock64(); pt[5] += t - t_last; t_last = t; }
        if (phase == PH_INIT) init_task(d, tl, i, base, snk, dist, bm, lane);
        else if (phase == PH_RELAX) relax_visit(d, tl, ldg(list + i), base, BP, rc, dist, rmask, S.relax, bm, lane);
        else if (phase == PH_SCAN) scan_task(d, tl, i, base, ex, dist, bm, &c->active, lane);
        else push_visit(d, tl, ldg(list + i), inner, base, BP, rc, ex, snk, dist, rmask, S.push, bm, lane);
        drain();
        int dn = 0;
        if (lane == 0) dn = atomicAdd(&c->done, 1) + 1;
        dn = __builtin_amdgcn_readfirstlane(dn);
        if (PROF) { const long long t = wall_clock64(); pt[phase] += t - t_last; pt[6 + phase] += 1; t_last = t; }
        if (dn == n) {
            finish_pass(tl, sc, c, G, bm, list, epoch, phase, my_x, err_flag, lane);
            if (PROF) { const long long t = wall_clock64(); pt[4] += t - t_last; t_last = t; }
        }
        return true;
    };
    int cur = -1;                                           // the image that fed this wave last: its pass probably has more
    unsigned rot = (blockIdx.x * WG_WAVES + wv) * 2654435761u >> 16;    // where in a chunk this wave looks first (spreads the herd)
    for (;;) {
        if (ldg(&G->finished) >= ldg(&G->n_open) || ldg(&G->error)) break;
        bool worked = cur >= 0 && try_take(cur);
        if (!worked) cur = -1;
        for (int cc = 0; cc < n_chunks && !worked; ++cc) {
            const int chunk = (chunk0 + cc) % n_chunks;
            const int bl = chunk * 64 + lane;
            // ---- which of my XCD's images has a task left in its current pass?  (a peek: the grab decides)
            bool avail = false;
            if (bl < d.B && ldg(&ctl[bl].owner) == my_x) {
                const unsigned long long h = ldg64(&ctl[bl].head), nl = ldg64(&ctl[bl].nlist);
                avail = (h >> 32) == (nl >> 32) && (unsigned)h < (unsigned)(nl & 0xfffff);
            }
            unsigned long long m = __ballot(avail);
            for (int tries = 0; m && !worked && tries < 3; ++tries) {
                const int sh = rot & 63;
                const unsigned long long mr = (m >> sh) | (sh ? m << (64 - sh) : 0ull);
                const int bit = (__ffsll((long long)mr) - 1 + sh) & 63;
                m &= ~(1ull << bit);
                rot += 7;
                if (try_take(chunk * 64 + bit)) { worked = true; cur = chunk * 64 + bit; chunk0 = chunk; }
            }
        }
        if (worked) { idle = 0; continue; }
        // ---- nothing to do for my XCD: adopt an unclaimed image (bounded per XCD, so that every XCD gets its share)
        bool claimed = false;
        for (int cc = 0; cc < n_chunks && !claimed; ++cc) {
            const int chunk = (chunk0 + cc) % n_chunks;
            const int bl = chunk * 64 + lane;
            const unsigned long long fm = __ballot(bl < d.B && ldg(&ctl[bl].owner) == OWNER_FREE);
            if (!fm) continue;
            int ok = 0;
            if (lane == 0) {
                // take a slot of my XCD only if one is free (an add-then-undo would keep the count above the cap while
                // hundreds of waves probe it), then the image; give the slot back if another XCD was faster
                const int cnt = ldg(&G->xcd_count[my_x]);
                if (cnt < cap && atomicCAS(&G->xcd_count[my_x], cnt, cnt + 1) == cnt) {
                    // spread the XCDs over the free images of the chunk
                    int pick = __ffsll((long long)fm) - 1;
                    const unsigned long long rot = fm >> ((my_x * 8) & 63);
                    if (rot) pick = ((my_x * 8) & 63) + __ffsll((long long)rot) - 1;
                    ok = atomicCAS(&ctl[chunk * 64 + pick].owner, OWNER_FREE, my_x) == OWNER_FREE;
                    if (!ok) atomicSub(&G->xcd_count[my_x], 1);
                }
            }
            claimed = __builtin_amdgcn_readfirstlane(ok) != 0;
            if (claimed) chunk0 = chunk;
            break;                                          // one attempt per idle turn
        }
        if (claimed) { idle = 0; continue; }
        for (long long k = 0; k <= (idle < 8 ? idle : 8); ++k) __builtin_amdgcn_s_sleep(32);      // back off: idle waves must not crowd the L2
        if (++idle > sc.max_idle) { if (lane == 0) { atomicExch(&G->error, 2); atomicOr(err_flag, 2); } break; }    // seconds without progress
    }
    if (PROF && lane == 0) {
        pt[5] += wall_clock64() - t_last;
#pragma unroll
        for (int k = 0; k < 10; ++k) atomicAdd(&prof[my_x * 16 + k], (unsigned long long)pt[k]);
        atomicAdd(&prof[my_x * 16 + 10], 1ull);
    }
}

int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::max(1, std::atoi(e)) : dflt;
}

} // namespace

bool maxflow_pool_fits(const GcDims& d) {
    return (int64_t)cdiv(d.W, PT_W) * cdiv(d.H, PT_H) <= MAX_TILES && d.W >= 1 && d.H >= 1;
}

int maxflow_pool(ggc_ctx* ctx, hipStream_t st, const GcDims& d, const int32_t* state, int32_t* rc, int32_t* ex,
                 int32_t* snk, int32_t* dist, uint8_t* rmask, int32_t* err_flag) {
    if (!maxflow_pool_fits(d)) return set_err(ctx, GGC_E_UNSUPPORTED, "image has more push tiles than the pooled max-flow's bitmap");
    const MfTiles tl{cdiv(d.W, RT), cdiv(d.H, RT), cdiv(d.W, PT_W), cdiv(d.H, PT_H)};
    const int stride = tl.pt_x * tl.pt_y;                 // >= relabel tiles per image
    int32_t* lists = scratch_t<int32_t>(ctx, S_GC_M, (size_t)d.B * stride);
    // control blocks | global block | bitmaps
    const size_t ctl_bytes = sizeof(PoolCtl) * d.B, g_bytes = 128, bm_bytes = sizeof(uint32_t) * BM_WORDS * (size_t)d.B;
    unsigned char* blk = scratch_t<unsigned char>(ctx, S_GC_N, ctl_bytes + g_bytes + bm_bytes);
    if (!lists || !blk) return GGC_E_OOM;
    PoolCtl* ctl = reinterpret_cast<PoolCtl*>(blk);
    PoolGlobal* G = reinterpret_cast<PoolGlobal*>(blk + ctl_bytes);
    uint32_t* bms = reinterpret_cast<uint32_t*>(blk + ctl_bytes + g_bytes);
    static DeviceOnce attr_done;
    if (attr_done.need(ctx->device)) {
        GGC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mf_pool<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(sizeof(WaveLds) * WG_WAVES)));
        GGC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mf_pool<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(sizeof(WaveLds) * WG_WAVES)));
        attr_done.done(ctx->device);
    }
    static const Sched sc{env_int("GGC_MFP_PASSES0", 6), env_int("GGC_MFP_PASSES", 8), env_int("GGC_MFP_INNER", 8),
                          env_int("GGC_MFP_TAIL_ACTIVE", 256), env_int("GGC_MFP_TAIL_PASSES", 16),
                          env_int("GGC_MFP_TAIL_INNER", 32), 4096, env_int("GGC_MFP_MAX_IDLE", 1 << 20)};
    static const int wg_per_cu = std::min(WG_PER_CU, env_int("GGC_MFP_WG_PER_CU", WG_PER_CU));
    GGC_HIP(ctx, hipMemsetAsync(blk + ctl_bytes, 0, g_bytes + bm_bytes, st));
    hipLaunchKernelGGL(k_pool_init, dim3(cdiv(d.B, 256)), dim3(256), 0, st, d.B, tl.rt_x * tl.rt_y, state, ctl, G);
    // an XCD adopts images while it owns fewer than its share of unfinished ones
    const int cap = std::max(1, cdiv(d.B, 8));
    static const bool trace = std::getenv("GGC_MF_TRACE") != nullptr;
    unsigned long long* prof_dev = nullptr;
    if (trace) {
        prof_dev = scratch_t<unsigned long long>(ctx, S_GC_K, 16 * 16);
        if (!prof_dev) return GGC_E_OOM;
        GGC_HIP(ctx, hipMemsetAsync(prof_dev, 0, sizeof(unsigned long long) * 256, st));
    }
    {
        ProfScope prof(ctx, st, "maxflow_pool");
        if (trace)
            hipLaunchKernelGGL(k_mf_pool<true>, dim3(ctx->n_cu * wg_per_cu), dim3(WG_T), sizeof(WaveLds) * WG_WAVES, st, d, tl, sc, cap, rc, ex,
                               snk, dist, rmask, ctl, G, bms, lists, stride, err_flag, prof_dev);
        else
            hipLaunchKernelGGL(k_mf_pool<false>, dim3(ctx->n_cu * wg_per_cu), dim3(WG_T), sizeof(WaveLds) * WG_WAVES, st, d, tl, sc, cap, rc, ex,
                               snk, dist, rmask, ctl, G, bms, lists, stride, err_flag, prof_dev);
    }
    GGC_LAUNCH_CHECK(ctx);
    if (trace) {                                           // diagnostics: the control blocks after the launch
        {
            std::vector<unsigned long long> hp(256);
            GGC_HIP(ctx, hipStreamSynchronize(st));
            GGC_HIP(ctx, hipMemcpy(hp.data(), prof_dev, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost));
            unsigned long long t[11] = {};
            for (int x = 0; x < 16; ++x) for (int k = 0; k < 11; ++k) t[k] += hp[x * 16 + k];
            const double w = (double)std::max<unsigned long long>(1, t[10]), us = 0.01;
            std::fprintf(stderr, "[ggc maxflow pool] %llu waves; per wave: init %.0f us (%.1f tasks), relax %.0f us (%.1f), scan %.0f us (%.1f), push %.0f us (%.1f), "
                         "transitions %.0f us, without a task %.0f us\n", t[10], us * t[0] / w, t[6] / w, us * t[1] / w, t[7] / w, us * t[2] / w, t[8] / w,
                         us * t[3] / w, t[9] / w, us * t[4] / w, us * t[5] / w);
            std::fprintf(stderr, "    per task: init %.1f us, relax %.1f us, scan %.1f us, push %.1f us;  waves per XCD:", us * t[0] / std::max<double>(1, t[6]),
                         us * t[1] / std::max<double>(1, t[7]), us * t[2] / std::max<double>(1, t[8]), us * t[3] / std::max<double>(1, t[9]));
            for (int x = 0; x < 8; ++x) std::fprintf(stderr, " %llu", hp[x * 16 + 10]);
            std::fprintf(stderr, "\n");
        }
        std::vector<PoolCtl> h(d.B);
        PoolGlobal hg;
        GGC_HIP(ctx, hipStreamSynchronize(st));
        GGC_HIP(ctx, hipMemcpy(h.data(), ctl, ctl_bytes, hipMemcpyDeviceToHost));
        GGC_HIP(ctx, hipMemcpy(&hg, G, sizeof(hg), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[ggc maxflow pool] B=%d open=%d finished=%d error=%d xcd_count=", d.B, hg.n_open, hg.finished, hg.error);
        for (int i = 0; i < 8; ++i) std::fprintf(stderr, "%d ", hg.xcd_count[i]);
        std::fprintf(stderr, "\n");
        int shown = 0;
        for (int b = 0; b < d.B && shown < 8; ++b)
            if (h[b].owner != OWNER_DONE || d.B <= 4) {
                ++shown;
                std::fprintf(stderr, "    image %d: owner %d head {%u,%u} nlist {epoch %u phase %d inner %d n %d} done %d active %d round %d passes_left %d\n", b,
                             h[b].owner, (unsigned)(h[b].head >> 32), (unsigned)h[b].head, (unsigned)(h[b].nlist >> 32), (int)((h[b].nlist >> 28) & 15),
                             (int)((h[b].nlist >> 20) & 255), (int)(h[b].nlist & 0xfffff), h[b].done, h[b].active, h[b].round, h[b].passes_left);
            }
    }
    return GGC_OK;
}

} // namespace ggc
