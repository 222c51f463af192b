// ggc_mf_sweep.h — in-register relabel sweeps over a 32x32 tile of labels + halo in LDS, shared by the max-flow drivers
// (ggc_maxflow.hip: work lists of the whole batch; ggc_maxflow_image.hip: one workgroup per image).
#pragma once
#include "ggc_gc.h"

namespace ggc {

__device__ __forceinline__ void mf_wave_sync() {     // LDS traffic of one wave is in order: only the compiler needs telling
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int mf_wave_or(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}

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
// Five neighbours instead of eight: a pass that walks in one direction looks at the three neighbours it comes from and the two
// beside it; the opposite pass of the same sweep looks at the other three and the same two.  Together they cover all eight arcs,
// so a sweep that changes nothing is still a fixpoint test, and the fixpoint (the exact distances) is the same — at 10 gated
// minima per pixel and sweep instead of 16 (the sweeps are bound by instruction issue: ~1 100 instructions each before).
// bA..bE: bit offsets of the five arcs inside the pixel's mask byte.
template <int bA, int bB, int bC, int bD, int bE>
__device__ __forceinline__ int relax_px5(int c, uint32_t inv, int pos, int vA, int vB, int vC, int vD, int vE) {
    const int nd = min3i(min3i(gated(vA, inv, pos + bA), gated(vB, inv, pos + bB), gated(vC, inv, pos + bC)),
                         gated(vD, inv, pos + bD), gated(vE, inv, pos + bE));
    return min(c, nd + 1);
}


// Arc masks of a 32x32 relabel tile come from rmask (1 byte per pixel, kept current by the push visits for the arcs inside
// their 32x8 tile).  An arc that LEAVES its pixel's push tile may have been re-opened by a push from the neighbouring tile
// after the owner wrote the mask, so those bits are taken from the capacities themselves: rows with y % 8 == 0 / 7 (arcs
// up / down: rows 0, 8 / 7, 15 of the lane's V-sweep segment) and columns 0 / 31 (arcs left / right: the lane's H-sweep
// pixel of that column).  15 loads per lane.  Masks are INVERTED (bit set = no arc).
struct MfBorderArcs {
    int fr[4][3], fc[3];
    __device__ __forceinline__ void load(const GcDims& d, const int32_t* __restrict__ rc, size_t BP, size_t base, int ty0, int tx0,
                                         int lx, int h) {
        const size_t cx = min(tx0 + lx, d.W - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // q: rows 0, 7, 8, 15 of the lane's segment
            const int r = (q >> 1) * 8 + ((q & 1) ? 7 : 0);
            const size_t i = base + (size_t)min(ty0 + 16 * h + r, d.H - 1) * d.W + cx;
#pragma unroll
            for (int t = 0; t < 3; ++t) fr[q][t] = rc[rc_idx(((q & 1) ? 3 + 2 * t : 2 + 2 * t), i)];   // 3,5,7 | 2,4,6
        }
        const size_t i = base + (size_t)min(ty0 + lx, d.H - 1) * d.W + min(tx0 + (h ? 31 : 0), d.W - 1);   // H-sweep row lx
        fc[0] = rc[rc_idx((h ? 1 : 0), i)];
        fc[1] = rc[rc_idx((h ? 5 : 4), i)];
        fc[2] = rc[rc_idx((h ? 6 : 7), i)];
    }
    // inverted mask of row r (0..15) of the lane's segment
    __device__ __forceinline__ uint32_t row(uint32_t m, int r) const {
        if ((r & 7) == 0 || (r & 7) == 7) {
            const int q = (r >> 3) * 2 + ((r & 7) ? 1 : 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const uint32_t bit = 1u << (((r & 7) ? 3 : 2) + 2 * t);
                m = (fr[q][t] > 0) ? (m & ~bit) : (m | bit);
            }
        }
        return m;
    }
    // inverted mask of the pixel (row lx, column h ? 31 : 0)
    __device__ __forceinline__ uint32_t col(uint32_t m, int h) const {
        const uint32_t b0 = 1u << (h ? 1 : 0), b1 = 1u << (h ? 5 : 4), b2 = 1u << (h ? 6 : 7);
        m = (fc[0] > 0) ? (m & ~b0) : (m | b0);
        m = (fc[1] > 0) ? (m & ~b1) : (m | b1);
        m = (fc[2] > 0) ? (m & ~b2) : (m | b2);
        return m;
    }
};

// A push tile is DIRTY when a neighbouring tile pushed into it since its masks were last exact (k_build_graph, or a relabel
// visit that re-read its border arcs): only then can a border bit of rmask be stale, and only then does a relabel visit pay
// for the 15 border loads — 4 us of its 10 us load phase, the three column loads being 64 scattered lines each.  The visit
// that re-reads them writes the corrected bytes back and clears the flags (relabel and push phases alternate, nothing moves
// capacities during a relabel), so later relabels of the solve take the short path until the next push across that edge.
// dirty: one word per 32x8 push tile.  Returns whether any of the relabel tile's (up to) 4 push tiles is dirty (wave-uniform).
__device__ __forceinline__ bool mf_tile_dirty(const int32_t* __restrict__ dirty, const MfTiles& tl, int b, int tyi, int txi, int lane) {
    const int py = tyi * (MF_RT / MF_PT_H) + (lane & 3);
    const int v = (lane < 4 && py < tl.pt_y) ? dirty[(size_t)b * tl.pt_x * tl.pt_y + (size_t)py * tl.pt_x + txi] : 0;
    return __any(v != 0);
}
// after the row and column patches: the corrected mask bytes of the border pixels go back to rmask, the flags are cleared
template <class RelaxTile>
__device__ __forceinline__ void mf_tile_repair(const GcDims& d, const MfTiles& tl, uint8_t* __restrict__ rmask, int32_t* __restrict__ dirty,
                                               const RelaxTile& S, size_t base, int b, int tyi, int txi, int ty0, int tx0, int lx, int h, int lane) {
    const uint8_t* sm = reinterpret_cast<const uint8_t*>(&S.m[0][0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (q >> 1) * 8 + ((q & 1) ? 7 : 0);
        const int y = ty0 + 16 * h + r, x = tx0 + lx;
        if (x < d.W && y < d.H) rmask[base + (size_t)y * d.W + x] = (uint8_t)(~sm[(16 * h + r) * MF_RT + lx] & 0xffu);
    }
    {
        const int y = ty0 + lx, x = tx0 + (h ? 31 : 0);
        if (x < d.W && y < d.H) rmask[base + (size_t)y * d.W + x] = (uint8_t)(~sm[lx * MF_RT + (h ? 31 : 0)] & 0xffu);
    }
    const int py = tyi * (MF_RT / MF_PT_H) + (lane & 3);
    if (lane < 4 && py < tl.pt_y) dirty[(size_t)b * tl.pt_x * tl.pt_y + (size_t)py * tl.pt_x + txi] = 0;
}

// lane = (column lx, half h): pixels (rows 16h .. 16h+15, column lx).  Returns 1 when a label changed.
template <class RelaxTile>
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
    for (int r = 0; r < 16; ++r) {                          // downwards: left, right, up, up-left, up-right (bits 0, 1, 2, 4, 6)
        const int a = r + 1;
        const int nv = relax_px5<0, 1, 2, 4, 6>(w[a][1], inv[r >> 2], 8 * (r & 3), w[a][0], w[a][2], w[a - 1][1], w[a - 1][0], w[a - 1][2]);
        chg |= (nv != w[a][1]) ? 1u << r : 0u;
        w[a][1] = nv;
    }
#pragma unroll
    for (int r = 15; r >= 0; --r) {                         // upwards: left, right, down, down-right, down-left (bits 0, 1, 3, 5, 7)
        const int a = r + 1;
        const int nv = relax_px5<0, 1, 3, 5, 7>(w[a][1], inv[r >> 2], 8 * (r & 3), w[a][0], w[a][2], w[a + 1][1], w[a + 1][2], w[a + 1][0]);
        chg |= (nv != w[a][1]) ? 1u << r : 0u;
        w[a][1] = nv;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((chg >> r) & 1u) S.d[16 * h + r + 1][lx + 1] = w[r + 1][1];
    return chg != 0u;
}
// lane = (row ly, half h): pixels (row ly, columns 16h .. 16h+15)
template <class RelaxTile>
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
    for (int k = 0; k < 16; ++k) {                          // rightwards: up, down, left, up-left, down-left (bits 2, 3, 0, 4, 7)
        const int c = k + 1;
        const int nv = relax_px5<2, 3, 0, 4, 7>(w[1][c], inv[k >> 2], 8 * (k & 3), w[0][c], w[2][c], w[1][c - 1], w[0][c - 1], w[2][c - 1]);
        chg |= (nv != w[1][c]) ? 1u << k : 0u;
        w[1][c] = nv;
    }
#pragma unroll
    for (int k = 15; k >= 0; --k) {                         // leftwards: up, down, right, up-right, down-right (bits 2, 3, 1, 6, 5)
        const int c = k + 1;
        const int nv = relax_px5<2, 3, 1, 6, 5>(w[1][c], inv[k >> 2], 8 * (k & 3), w[0][c], w[2][c], w[1][c + 1], w[0][c + 1], w[2][c + 1]);
        chg |= (nv != w[1][c]) ? 1u << k : 0u;
        w[1][c] = nv;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if ((chg >> k) & 1u) S.d[ly + 1][16 * h + k + 1] = w[1][k + 1];
    return chg != 0u;
}

} // namespace ggc
