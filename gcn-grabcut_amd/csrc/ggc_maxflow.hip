// ggc_maxflow.hip — max-flow / min-cut for GrabCut (C5: GCGraph::maxFlow + the
// source/sink labelling of estimateSegmentation) on the
// implicit 8-neighbour pixel grid with int32 capacities, all images of a batch at once.
//
// Algorithm: lock-free push-relabel (Hong 2008) in rounds of
//     global relabel  (exact BFS distances to the sink, as a min-plus relaxation)
//     -> push-relabel sweeps
// until no pixel holds excess that can still reach the sink.  Only phase 1
// (maximum preflow) is needed: a pixel is foreground iff it cannot reach the sink
// in the residual graph, which the last relabel has just computed.
//
// Mapping to the hardware:
//  * both phases work on LDS-resident tiles.  Relabel: a 32x32 tile + halo of labels
//    is relaxed to a local fixpoint per visit.  Push: a 32x8 tile of excess / residual
//    capacities / labels runs `inner` sweeps with LDS atomics; pushes across the tile
//    edge use global atomics, and the write-back applies DELTAS atomically because a
//    neighbouring tile may have added to this tile's excess or reverse arcs meanwhile.
//    Every stale value is a lower bound of the true one, which is the asynchrony the
//    lock-free algorithm tolerates.
//  * activity is sparse (about 1 % of the pixels after the first round, a handful of
//    images in the last rounds), so every launch walks a WORK LIST of tiles instead of
//    the whole grid: a tile that changes (relabel) or keeps / hands over excess (push)
//    appends itself or its neighbour to the next list (flag + atomic counter), and a
//    launch costs a few microseconds when there is little to do.  Three rotating
//    counters let the launch that consumes list L also clear the counter of list L+2.
//  * images without active pixels leave the open-image list; converged tiles cost nothing.
#include "ggc_gc.h"
#include "ggc_mf_sweep.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

namespace ggc {

constexpr int RT = MF_RT;              // relabel tile side
constexpr int PT_W = MF_PT_W, PT_H = MF_PT_H;   // push tile (32x16 measured 6 % slower end to end)
constexpr int PT_N = PT_W * PT_H;      // threads of a push block, one pixel each
// blocks per work-list launch and 64 open images: a block walks several tiles of the list (measured best, tools/mf_knobs.sh)
constexpr int PUSH_GRID = 1024, RELAX_GRID = 1024, ASYNC_GRID = 128;

__device__ __forceinline__ int ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// append `tile` to the next work list unless it is already on it
__device__ __forceinline__ void push_tile(int tile, int32_t* __restrict__ flag, int32_t* __restrict__ list,
                                          int32_t* __restrict__ count) {
    if (ld(&flag[tile]) == 0 && atomicExch(&flag[tile], 1) == 0) list[atomicAdd(count, 1)] = tile;   // cheap test first
}

// up to three int32 regions zeroed by ONE launch (a round used to issue seven hipMemsetAsync calls: each is a launch of its own)
__global__ void __launch_bounds__(256) k_mf_zero3(int32_t* __restrict__ a, size_t na, int32_t* __restrict__ b, size_t nb, int32_t* __restrict__ c, size_t nc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < na) a[i] = 0;
    else if (i < na + nb) b[i - na] = 0;
    else if (i < na + nb + nc) c[i - na - nb] = 0;
}
void mf_zero3(hipStream_t st, int32_t* a, size_t na, int32_t* b, size_t nb, int32_t* c, size_t nc) {
    const size_t n = na + nb + nc;
    if (n) hipLaunchKernelGGL(k_mf_zero3, dim3(cdiv(n, 256)), dim3(256), 0, st, a, na, b, nb, c, nc);
}

// The same through a per-block list.  Every append used to be a returning atomicAdd on the ONE counter of the next list, and
// same-address atomics retire at ~90 per microsecond chip-wide: a dense launch appends ~15 000 tiles, i.e. >= 150 us of
// counter traffic — which is what those launches took (139-165 us), whatever the grid or the occupancy.  A block now
// collects its tiles in LDS and reserves their slots with one atomicAdd when it is done.
constexpr int OUT_CAP = 1024;
struct OutList { int n, base; int buf[OUT_CAP]; };
__device__ __forceinline__ void push_tile_l(int tile, int32_t* __restrict__ flag, OutList& L, int32_t* __restrict__ list,
                                            int32_t* __restrict__ count) {
    if (ld(&flag[tile]) == 0 && atomicExch(&flag[tile], 1) == 0) {
        const int i = atomicAdd(&L.n, 1);
        if (i < OUT_CAP) L.buf[i] = tile;
        else list[atomicAdd(count, 1)] = tile;                             // (a block that overflows its list appends directly)
    }
}
// all threads of the block, once, after their last push_tile_l
__device__ __forceinline__ void flush_tiles(OutList& L, int32_t* __restrict__ list, int32_t* __restrict__ count) {
    __syncthreads();
    const int n = min(L.n, OUT_CAP);
    if (threadIdx.x == 0 && n > 0) L.base = atomicAdd(count, n);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) list[L.base + i] = L.buf[i];
}

// Start of a global relabel by TILES (the wave-per-tile relabel kernels): a wave writes the starting labels of one 32x32
// tile (1 next to the sink, infinity elsewhere) and puts the tile on the first work list only if it holds a pixel WITHOUT a
// sink link — a tile whose pixels all touch the sink (59 % of the bench's pixels are definite background) is final at
// label 1 and never needs a visit.  (Folding this pass into the first relabel launch was measured in round 3: the heavy kernel then visits every tile, 99 us against 18 + 70.)
// It is the first kernel of a round, so it also leaves the round's bookkeeping clean (each used to be a zero-fill launch of
// its own): BOTH membership flags of every relabel tile it looks at (tiles of closed images are never listed again), the
// image's active-pixel count, and the words the active scan and k_done_update accumulate into (scan[0]: open images,
// scan[4..6]: push-list counters, scan[7]: active total).  The relabel counters themselves are zeroed by k_done_update / k_open_init.
__global__ void __launch_bounds__(256) k_mf_rinit(GcDims d, MfTiles tl, const int32_t* __restrict__ open_list,
                                                  const int32_t* __restrict__ snk, int32_t* __restrict__ dist,
                                                  int32_t* __restrict__ list, int32_t* __restrict__ flag, int32_t* __restrict__ flag2,
                                                  int32_t* __restrict__ count, int32_t* __restrict__ active, int32_t* __restrict__ scan) {
    constexpr int T = MF_RT;
    __shared__ OutList outl;
    if (threadIdx.x == 0) outl.n = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) active[open_list[blockIdx.y]] = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8 && (threadIdx.x == 0 || threadIdx.x >= 4)) scan[threadIdx.x] = 0;   // not [1..3]: the relabel counters this launch appends to
    __syncthreads();
    const int tiles_per_image = tl.rt_x * tl.rt_y;
    const int tr = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tr < tiles_per_image) {
        const int b = open_list[blockIdx.y];
        const int ty0 = (tr / tl.rt_x) * T, tx0 = (tr % tl.rt_x) * T;
        const int lx = lane & 31, h = lane >> 5;
        const size_t base = (size_t)b * d.P;
        int sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = snk[base + (size_t)min(ty0 + 16 * h + r, d.H - 1) * d.W + min(tx0 + lx, d.W - 1)];
        bool open_px = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int y = ty0 + 16 * h + r, x = tx0 + lx;
            if (x < d.W && y < d.H) {
                dist[base + (size_t)y * d.W + x] = sv[r] > 0 ? 1 : DINF;
                open_px |= sv[r] <= 0;
            }
        }
        const bool listed = __any(open_px);
        if (lane == 0) {
            const int tile = b * tiles_per_image + tr;
            flag[tile] = listed ? 1 : 0;                                   // one writer per tile
            flag2[tile] = 0;
            if (listed) {
                const int i = atomicAdd(&outl.n, 1);
                if (i < OUT_CAP) outl.buf[i] = tile; else list[atomicAdd(count, 1)] = tile;
            }
        }
    }
    flush_tiles(outl, list, count);
}

// Global relabel over a work list of 32x32 tiles, a WAVE per tile (4 tiles per block, no block barrier): the labels are relaxed by alternating
// vertical and horizontal in-register sweeps (ggc_mf_sweep.h), which carry a front across the tile in a handful of
// sweeps where the neighbour-at-a-time iteration above needs one per pixel of the way.  32x32 tiles only.
struct RelaxWaveLds { int d[MF_RT + 2][MF_RT + 2]; uint32_t m[MF_RT][MF_RT / 4]; };
template <bool PROF>
__global__ void __launch_bounds__(256) k_mf_relax_wave(GcDims d, MfTiles tl, int phase, long long* __restrict__ prof, uint8_t* __restrict__ rmask,
                                                       int32_t* __restrict__ dirty, const int32_t* __restrict__ rc, int32_t* __restrict__ dist, int32_t* __restrict__ counters,
                                                       const int32_t* __restrict__ list_in, int32_t* __restrict__ list_out,
                                                       int32_t* __restrict__ flag_in, int32_t* __restrict__ flag_out) {
    constexpr int T = MF_RT, N_HALO = (T + 2) * (T + 2), HALO_IT = (N_HALO + 63) / 64;
    __shared__ RelaxWaveLds lds[4];
    __shared__ OutList outl;
    if (threadIdx.x == 0) outl.n = 0;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    RelaxWaveLds& S = lds[wv];
    const int n_in = counters[phase % 3];
    int32_t* n_out = counters + (phase + 1) % 3;
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[(phase + 2) % 3] = 0;       // the list after next starts empty
    const int tiles_per_image = tl.rt_x * tl.rt_y;
    const int G = gridDim.x * 4;
    int* sd = &S.d[0][0];
    uint8_t* sm = reinterpret_cast<uint8_t*>(&S.m[0][0]);
    long long pa = 0, pb = 0, pc = 0, pn = 0, psw = 0;
    // the list entry of the NEXT visit is requested one visit ahead: read at the top of its own visit it is a memory round
    // trip in front of the tile's 50 loads (measured: load + fill 12 us of a 22 us visit)
    int tile_nx = blockIdx.x * 4 + wv < n_in ? list_in[blockIdx.x * 4 + wv] : 0;
    for (int t = blockIdx.x * 4 + wv; t < n_in; t += G) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));                                     // keeps the lane arithmetic inside the loop (no hoist + spill)
        const long long t_a = PROF ? wall_clock64() : 0;
        const int tile = __builtin_amdgcn_readfirstlane(tile_nx);
        tile_nx = t + G < n_in ? list_in[t + G] : 0;
        const int b = tile / tiles_per_image, tr = tile % tiles_per_image;
        const int tyi = tr / tl.rt_x, txi = tr % tl.rt_x;
        const int tx0 = txi * T, ty0 = tyi * T;
        const size_t base = (size_t)b * d.P, BP = (size_t)d.B * d.P;
        const int lx = lane & 31, h = lane >> 5;
        int hv[HALO_IT];
#pragma unroll
        for (int k = 0; k < HALO_IT; ++k) {                                // unconditional loads from clamped addresses
            const int i = min(lane + k * 64, N_HALO - 1);
            const int gy = ty0 + i / (T + 2) - 1, gx = tx0 + i % (T + 2) - 1;
            hv[k] = dist[base + (size_t)min(max(gy, 0), d.H - 1) * d.W + min(max(gx, 0), d.W - 1)];
        }
        uint32_t mv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mv[r] = rmask[base + (size_t)min(ty0 + 16 * h + r, d.H - 1) * d.W + min(tx0 + lx, d.W - 1)];
        const bool dirty_t = mf_tile_dirty(dirty, tl, b, tyi, txi, lane);  // wave-uniform: a neighbour pushed into one of its push tiles
        MfBorderArcs ba;
        if (dirty_t) ba.load(d, rc, BP, base, ty0, tx0, lx, h);
        if (lane == 0) flag_in[tile] = 0;                                  // consumed
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
            if (i < N_HALO) sd[i] = (gx >= 0 && gx < d.W && gy >= 0 && gy < d.H) ? hv[k] : DINF;
        }
        mf_wave_sync();
        if (dirty_t) {
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
        const long long t_b = PROF ? wall_clock64() : 0;
        int n_sw = 0;
        for (int it = 0; it < 4 * T; ++it) {                               // a sweep pair that changes nothing: fixpoint
            const int ch = (it & 1) ? relax_sweep_h(S, inv_h, lx, h) : relax_sweep_v(S, inv_v, lx, h);
            mf_wave_sync();
            if (PROF) ++n_sw;
            if (!__any(ch)) { settled = true; break; }
        }
        const long long t_c = PROF ? wall_clock64() : 0;
        int nbm = settled ? 0 : 1 << 4;                                    // bit (dy + 1) * 3 + (dx + 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ly = 16 * h + r;
            const int v = S.d[ly + 1][lx + 1];
            if (v != old[r]) {
                dist[base + (size_t)(ty0 + ly) * d.W + tx0 + lx] = v;
                const int Lf = lx == 0, Rt = lx == T - 1, U = ly == 0, D = ly == T - 1;
                nbm |= (U & Lf) | U << 1 | (U & Rt) << 2 | Lf << 3 | Rt << 5 | (D & Lf) << 6 | D << 7 | (D & Rt) << 8;
            }
        }
        nbm = mf_wave_or(nbm);
        if (lane < 9 && (nbm >> lane) & 1) {
            const int ty = tyi + lane / 3 - 1, tx = txi + lane % 3 - 1;
            if (ty >= 0 && ty < tl.rt_y && tx >= 0 && tx < tl.rt_x)
                push_tile_l(b * tiles_per_image + ty * tl.rt_x + tx, flag_out, outl, list_out, n_out);
        }
        mf_wave_sync();
        if (PROF) { pa += t_b - t_a; pb += t_c - t_b; pc += wall_clock64() - t_c; pn += 1; psw += n_sw; }
    }
    if (PROF && (threadIdx.x & 63) == 0 && pn) {                           // GGC_MF_TRACE: visit-phase clocks, one set of atomics per wave
        unsigned long long* q = reinterpret_cast<unsigned long long*>(prof) + (64 + (blockIdx.x & 63)) * 8;
        atomicAdd(&q[0], (unsigned long long)pa); atomicAdd(&q[1], (unsigned long long)pb); atomicAdd(&q[2], (unsigned long long)pc);
        atomicAdd(&q[3], (unsigned long long)pn); atomicAdd(&q[4], (unsigned long long)psw);
    }
    flush_tiles(outl, list_out, n_out);
}

// Push-relabel sweeps over a work list of 32x8 tiles.  PPT pixels per thread: a dense launch is bound by tile visits in
// flight per CU (visit latency ~9 us x the blocks a CU holds), and a CU holds 32 waves whatever the block size, so
// PPT = 2 (two waves per tile) doubles the tiles in flight.
template <int PPT>
__global__ void __launch_bounds__(PT_N / PPT) k_mf_pr_list(GcDims d, MfTiles tl, int phase, int inner,
                                                          int32_t* __restrict__ rc, int32_t* __restrict__ ex,
                                                          int32_t* __restrict__ snk, int32_t* __restrict__ dist, uint8_t* __restrict__ rmask,
                                                          int32_t* __restrict__ dirty, int32_t* __restrict__ counters, const int32_t* __restrict__ list_in,
                                                          int32_t* __restrict__ list_out, int32_t* __restrict__ flag_in,
                                                          int32_t* __restrict__ flag_out) {
    constexpr int NT = PT_N / PPT;                                         // threads; pixel slot of (thread, j) = tid + j * NT
    __shared__ int s_ex[PT_N];
    __shared__ int s_d[PT_H + 2][PT_W + 2];
    __shared__ int s_rc[8][PT_N];
    __shared__ OutList outl;
    __shared__ int s_nbm;                                                  // bit (dy + 1) * 3 + (dx + 1): tiles to put on the next list
    const int tid = threadIdx.x, lx = tid & 31;
    if (tid == 0) outl.n = 0;                                              // (the first barrier of the tile loop orders it)
    const int n_in = counters[phase % 3];
    int32_t* n_out = counters + (phase + 1) % 3;
    if (blockIdx.x == 0 && tid == 0) counters[(phase + 2) % 3] = 0;
    const int tiles_per_image = tl.pt_x * tl.pt_y;
    const size_t BP = (size_t)d.B * d.P;
    // The visit latency bounds the dense launches, so the next tile of this block is loaded into registers while the
    // current one is swept (its list entry one step earlier still).  Safe for the same reason concurrent tiles are:
    // ring pixels are written back as atomic deltas, interior pixels have no other writer during a launch.
    constexpr int N_HALO = (PT_H + 2) * (PT_W + 2), HALO_IT = (N_HALO + NT - 1) / NT;
    struct TileRegs { int e[PPT], sk[PPT], r[PPT][8], hv[HALO_IT]; };
    auto load_tile = [&](int tile, TileRegs& R) {
        const int b = tile / tiles_per_image, tr = tile % tiles_per_image;
        const int tyi = tr / tl.pt_x, txi = tr % tl.pt_x;
        const int x = txi * PT_W + lx;
        const size_t base = (size_t)b * d.P;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int y = tyi * PT_H + ((tid + j * NT) >> 5);
            const bool in = x < d.W && y < d.H;
            const int p = y * d.W + x;
            R.e[j] = in ? ex[base + p] : 0;
            R.sk[j] = in ? snk[base + p] : 0;
#pragma unroll
            for (int dir = 0; dir < 8; ++dir) R.r[j][dir] = in ? rc[rc_idx(dir, base + p)] : 0;
        }
#pragma unroll
        for (int k = 0; k < HALO_IT; ++k) {
            const int i = tid + k * NT;
            const int gy = tyi * PT_H + i / (PT_W + 2) - 1, gx = txi * PT_W + i % (PT_W + 2) - 1;
            R.hv[k] = (i < N_HALO && gx >= 0 && gx < d.W && gy >= 0 && gy < d.H) ? dist[base + (size_t)gy * d.W + gx] : DINF;
        }
    };
    const int G = gridDim.x;
    int t = blockIdx.x;
    if (t >= n_in) return;
    int tile_next = list_in[t];
    int tile_next2 = t + G < n_in ? list_in[t + G] : 0;
    TileRegs N;
    load_tile(tile_next, N);
    for (; t < n_in; t += G) {
        const int tile = tile_next;
        const int b = tile / tiles_per_image, tr = tile % tiles_per_image;
        const int tyi = tr / tl.pt_x, txi = tr % tl.pt_x;
        const int x = txi * PT_W + lx;
        const size_t base = (size_t)b * d.P;
        __syncthreads();
        int e0[PPT], sk0[PPT], sk[PPT], d0[PPT], r0[PPT][8], pp[PPT];
        bool inb[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int slot = tid + j * NT, y = tyi * PT_H + (slot >> 5);
            inb[j] = x < d.W && y < d.H;
            pp[j] = y * d.W + x;
            e0[j] = N.e[j]; sk0[j] = N.sk[j]; sk[j] = sk0[j];
            s_ex[slot] = e0[j];
#pragma unroll
            for (int dir = 0; dir < 8; ++dir) { r0[j][dir] = N.r[j][dir]; s_rc[dir][slot] = r0[j][dir]; }
        }
#pragma unroll
        for (int k = 0; k < HALO_IT; ++k) {
            const int i = tid + k * NT;
            if (i < N_HALO) s_d[i / (PT_W + 2)][i % (PT_W + 2)] = N.hv[k];
        }
        if (t + G < n_in) {                                                // block-uniform
            tile_next = tile_next2;
            tile_next2 = t + 2 * G < n_in ? list_in[t + 2 * G] : 0;
            load_tile(tile_next, N);
        }
        __syncthreads();
        if (tid == 0) { flag_in[tile] = 0; s_nbm = 0; }                                   // consumed
#pragma unroll
        for (int j = 0; j < PPT; ++j) d0[j] = s_d[((tid + j * NT) >> 5) + 1][lx + 1];
        int nbm = 0;
        for (int it = 0; it < inner; ++it) {
            int act = 0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                if (!inb[j]) continue;
                const int slot = tid + j * NT, ly = slot >> 5;
                const int e = __hip_atomic_load(&s_ex[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int dp = s_d[ly + 1][lx + 1];
                if (e > 0 && dp < d.P) {
                    act = 1;
                    // the 8 residual capacities and 8 neighbour labels are read in one batch and the arg-min is branch-free:
                    // one LDS round trip per sweep instead of a chain of up to 16
                    int r[8], hq[8];
#pragma unroll
                    for (int dir = 0; dir < 8; ++dir) {
                        r[dir] = __hip_atomic_load(&s_rc[dir][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        hq[dir] = s_d[ly + 1 + dir_dy(dir)][lx + 1 + dir_dx(dir)];
                    }
                    int hmin = sk[j] > 0 ? 0 : DINF, best = sk[j] > 0 ? 8 : -1, rb = 0;
#pragma unroll
                    for (int dir = 0; dir < 8; ++dir) {
                        const bool ok = r[dir] > 0 && hq[dir] < hmin;
                        hmin = ok ? hq[dir] : hmin; best = ok ? dir : best; rb = ok ? r[dir] : rb;
                    }
                    if (best >= 0 && dp > hmin) {
                        if (best == 8) {
                            const int dl = min(e, sk[j]);
                            sk[j] -= dl;
                            atomicSub(&s_ex[slot], dl);
                        } else {
                            const int dl = min(e, rb);
                            atomicSub(&s_rc[best][slot], dl);
                            atomicSub(&s_ex[slot], dl);
                            const int bx = dir_dx(best), by = dir_dy(best);
                            const int qlx = lx + bx, qly = ly + by;
                            if (qlx >= 0 && qlx < PT_W && qly >= 0 && qly < PT_H) {
                                const int qt = qly * PT_W + qlx;
                                atomicAdd(&s_rc[best ^ 1][qt], dl);
                                atomicAdd(&s_ex[qt], dl);
                            } else {                                        // across the tile edge: straight to global memory
                                const int q = pp[j] + by * d.W + bx;
                                const int y = tyi * PT_H + ly;
                                atomicAdd(&rc[rc_idx((best ^ 1), base + q)], dl);
                                atomicAdd(&ex[base + q], dl);
                                // (the neighbour tile is told after the sweeps, see k_mf_pr_wave)
                                const int tdy = qly < 0 ? -1 : (qly >= PT_H ? 1 : 0), tdx = qlx < 0 ? -1 : (qlx >= PT_W ? 1 : 0);
                                nbm |= 1 << ((tdy + 1) * 3 + tdx + 1);
                            }
                        }
                    } else {
                        s_d[ly + 1][lx + 1] = (best >= 0 && hmin < DINF) ? hmin + 1 : DINF;
                    }
                }
            }
            if (!__syncthreads_or(act)) break;
        }
        int left = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            if (!inb[j]) continue;
            // only the tile's border ring can receive pushes from other tiles during this launch: the interior is
            // owned exclusively, so its write-back is a plain store (L2 atomics are the scarce resource in the
            // early rounds, when nearly every pixel changes)
            const int slot = tid + j * NT, ly = slot >> 5, p = pp[j];
            const bool ring = lx == 0 || lx == PT_W - 1 || ly == 0 || ly == PT_H - 1;
            const int e1 = s_ex[slot];
            if (e1 != e0[j]) { if (ring) atomicAdd(&ex[base + p], e1 - e0[j]); else ex[base + p] = e1; }
            int m1 = 0, chg = 0;
#pragma unroll
            for (int dir = 0; dir < 8; ++dir) {
                const int r1 = s_rc[dir][slot];
                m1 |= (r1 > 0) ? (1 << dir) : 0;
                if (r1 != r0[j][dir]) {
                    chg = 1;
                    // another tile's push can only add to an arc that points OUT of this tile (the reverse of its own arc):
                    // those go back as atomic deltas, every other capacity is this block's alone (L2 atomics, ~30 G/s chip-wide,
                    // are what bounds the dense rounds)
                    const bool out = (ly == 0 && dir_dy(dir) < 0) || (ly == PT_H - 1 && dir_dy(dir) > 0) ||
                                     (lx == 0 && dir_dx(dir) < 0) || (lx == PT_W - 1 && dir_dx(dir) > 0);
                    if (out) atomicAdd(&rc[rc_idx(dir, base + p)], r1 - r0[j][dir]);
                    else rc[rc_idx(dir, base + p)] = r1;
                }
            }
            if (chg) rmask[base + p] = (uint8_t)m1;     // (arcs that leave the tile: the relabel reads the capacities, see k_mf_dinit)
            if (sk[j] != sk0[j]) snk[base + p] = sk[j];
            const int d1 = s_d[ly + 1][lx + 1];
            if (d1 != d0[j]) dist[base + p] = d1;
            left |= (e1 > 0 && d1 < d.P) ? 1 : 0;
        }
        if (nbm) atomicOr(&s_nbm, nbm);
        if (__syncthreads_or(left) && tid == 0) s_nbm |= 1 << 4;           // still has work
        __syncthreads();
        if (tid < 9 && (s_nbm >> tid) & 1) {
            const int ty = tyi + tid / 3 - 1, tx = txi + tid % 3 - 1;
            if (ty >= 0 && ty < tl.pt_y && tx >= 0 && tx < tl.pt_x) {
                const int nbt = b * tiles_per_image + ty * tl.pt_x + tx;
                if (tid != 4) dirty[nbt] = 1;                              // its border arcs may have been re-opened (ggc_mf_sweep.h)
                push_tile_l(nbt, flag_out, outl, list_out, n_out);
            }
        }
    }
    flush_tiles(outl, list_out, n_out);
}

// per image: number of pixels whose excess can still reach the sink; their push tiles form the round's first work list.
// A wave scans one 32x8 push tile (4 pixels per lane, rows of 128 bytes) and appends it once when it holds an active pixel:
// no per-pixel membership test.  (The pixel-strided scan this replaces sent every active pixel through the tile's flag —
// 920 k flag reads and LDS appends in the first round of a 64-image solve: 203 us, against ~25 us for the 61 MB it reads.)
__global__ void __launch_bounds__(256) k_mf_active(GcDims d, MfTiles tl, const int32_t* __restrict__ open_list,
                                                   const int32_t* __restrict__ ex, const int32_t* __restrict__ dist,
                                                   int32_t* __restrict__ active, int32_t* __restrict__ flag, int32_t* __restrict__ flag2,
                                                   int32_t* __restrict__ list, int32_t* __restrict__ count) {
    __shared__ OutList outl;
    __shared__ int s_n;
    if (threadIdx.x == 0) { outl.n = 0; s_n = 0; }
    __syncthreads();
    const int tiles_per_image = tl.pt_x * tl.pt_y;
    const int b = open_list[blockIdx.y];
    const int lane = threadIdx.x & 63, lx = lane & 31, r0 = lane >> 5;
    const size_t base = (size_t)b * d.P;
    int n_block = 0;
    for (int tr = blockIdx.x * 4 + (threadIdx.x >> 6); tr < tiles_per_image; tr += gridDim.x * 4) {
        const int tyi = tr / tl.pt_x, txi = tr - tyi * tl.pt_x;
        const int x = txi * PT_W + lx;
        int ev[PT_H / 2], dv[PT_H / 2];
#pragma unroll
        for (int j = 0; j < PT_H / 2; ++j) {                               // every load issued from a clamped address, then masked
            const int y = tyi * PT_H + r0 + 2 * j;
            const size_t i = base + (size_t)min(y, d.H - 1) * d.W + min(x, d.W - 1);
            ev[j] = ex[i]; dv[j] = dist[i];
        }
        int n = 0;
#pragma unroll
        for (int j = 0; j < PT_H / 2; ++j) {
            const int y = tyi * PT_H + r0 + 2 * j;
            n += (x < d.W && y < d.H && ev[j] > 0 && dv[j] < DINF) ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
        if (lane == 0) {
            const int tile = b * tiles_per_image + tr;
            flag[tile] = n > 0 ? 1 : 0;                                    // both membership flags of every push tile of an open image: one writer
            flag2[tile] = 0;                                               // per tile, no zero-fill launch before the scan
            if (n > 0) {
                const int i = atomicAdd(&outl.n, 1);
                if (i < OUT_CAP) outl.buf[i] = tile; else list[atomicAdd(count, 1)] = tile;
                n_block += n;
            }
        }
    }
    if (lane == 0 && n_block) atomicAdd(&s_n, n_block);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&active[b], s_n);
    flush_tiles(outl, list, count);
}

// closes images without active pixels and compacts the still-open ones into the next launch list
__global__ void k_done_update(int n_cur, const int32_t* __restrict__ list_cur, const int32_t* __restrict__ active,
                              int32_t* __restrict__ list_nxt, int32_t* __restrict__ n_open, int keep_all, int32_t* __restrict__ rl_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3) rl_cnt[i] = 0;                              // this round's relabel is over: its work-list counters start the next one at zero
    if (i >= n_cur) return;
    const int b = list_cur[i];
    if (active[b] != 0 || keep_all) { list_nxt[atomicAdd(n_open, 1)] = b; atomicAdd(n_open + 7, active[b]); }   // [7]: active pixels in total
}
__global__ void k_open_init(int B, const int32_t* __restrict__ state, int32_t* __restrict__ list, int32_t* __restrict__ n_open,
                            int32_t* __restrict__ rl_cnt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < 3) rl_cnt[b] = 0;
    if (b < B && !state[b]) list[atomicAdd(n_open, 1)] = b;
}

// small synchronous device -> host read through the context's page-locked staging buffer.  (Round 3 measured the
// alternative — a one-thread kernel publishing into host-coherent memory and the host spinning on a ticket instead of
// hipStreamSynchronize: 58.4-58.8 vs 59.1-60.3 ms per GrabCut stage, 86.9 vs 87.1 ms per step: not worth four spinning cores.)
int read_i32(ggc_ctx* ctx, hipStream_t st, const int32_t* dev, int n, std::vector<int32_t>& host) {
    host.resize(n);
    if (ctx->h_pinned && n <= ggc_ctx::H_PINNED_INTS) {
        GGC_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, dev, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
        GGC_HIP(ctx, hipStreamSynchronize(st));
        std::copy(ctx->h_pinned, ctx->h_pinned + n, host.begin());
        return GGC_OK;
    }
    GGC_HIP(ctx, hipMemcpyAsync(host.data(), dev, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    GGC_HIP(ctx, hipStreamSynchronize(st));
    return GGC_OK;
}

int maxflow(ggc_ctx* ctx, hipStream_t st, const GcDims& d, const int32_t* state, int32_t* rc, int32_t* ex,
            int32_t* snk, int32_t* dist, uint8_t* rmask, int32_t* lists /*[2B] open-image lists*/,
            int32_t* flags /*[B] active | [1] n_open | [3] relabel counters | [3] push counters | [1] active total*/,
            int32_t* err_flag, bool masks_exact) {
    const Knobs& kn = knobs();
    const int B = d.B;
    int32_t* active = flags;
    int32_t* n_open = flags + B;
    int32_t* rl_cnt = flags + B + 1;
    int32_t* pr_cnt = flags + B + 4;
    const MfTiles tl{cdiv(d.W, RT), cdiv(d.H, RT), cdiv(d.W, PT_W), cdiv(d.H, PT_H)};
    const size_t n_rt = (size_t)tl.rt_x * tl.rt_y * B, n_pt = (size_t)tl.pt_x * tl.pt_y * B;
    // per tile kind: two ping-pong work lists and two ping-pong membership flags
    int32_t* rl = scratch_t<int32_t>(ctx, S_GC_N, n_rt * 4);
    int32_t* pt = scratch_t<int32_t>(ctx, S_GC_M, n_pt * 4);
    if (!rl || !pt) return GGC_E_OOM;
    int32_t *rl_list[2] = {rl, rl + n_rt}, *rl_flag[2] = {rl + 2 * n_rt, rl + 3 * n_rt};
    int32_t *pt_list[2] = {pt, pt + n_pt}, *pt_flag[2] = {pt + 2 * n_pt, pt + 3 * n_pt};
    int32_t *list_cur = lists, *list_nxt = lists + B;
    // asynchronous single-launch drivers of the sparse phases (ggc_maxflow_async.hip): ring of the larger tile count, the
    // queue words, one lock word per push tile; behind them the trace clocks and the dirty words
    const size_t ring_cap = std::max(n_rt, n_pt);
    GGC_REQUIRE(ctx, ring_cap < (1u << 24), GGC_E_UNSUPPORTED, "batch has more max-flow tiles than a queue entry addresses");
    unsigned long long* ring = scratch_t<unsigned long long>(ctx, S_GC_O, ring_cap + (AQ_WORDS + n_pt + 1) / 2 + 1 + 128 * 8 + 2 + (n_pt + 1) / 2 + 1);
    if (!ring) return GGC_E_OOM;
    int32_t* aq = reinterpret_cast<int32_t*>(ring + ring_cap);
    int32_t* busy = aq + AQ_WORDS;
    long long* prof_dev = kn.mf_trace ? reinterpret_cast<long long*>(busy + ((n_pt + 3) & ~(size_t)1)) : nullptr;
    // one word per push tile: a neighbour pushed into it since its arc masks were last exact.  k_build_graph writes exact
    // masks for every pixel on a cold start; a warm start leaves the definite pixels alone, so their tiles' marks stay.
    int32_t* dirty = busy + ((n_pt + 3) & ~(size_t)1) + 2 * 128 * 8;
    if (masks_exact) GGC_HIP(ctx, hipMemsetAsync(dirty, 0, sizeof(int32_t) * n_pt, st));
    if (prof_dev) GGC_HIP(ctx, hipMemsetAsync(prof_dev, 0, 128 * 8 * sizeof(long long), st));
    GGC_HIP(ctx, hipMemsetAsync(n_open, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_open_init, dim3(cdiv(B, 256)), dim3(256), 0, st, B, state, list_cur, n_open, rl_cnt);
    GGC_LAUNCH_CHECK(ctx);
    std::vector<int32_t> host;
    int rcode = read_i32(ctx, st, n_open, 1, host);
    if (rcode) return rcode;
    int n_cur = host[0];
    if (n_cur == 0) return GGC_OK;
    const int max_rounds = 4096;
    auto t_prev = std::chrono::steady_clock::now();
    double push_ms = 0.0;
    for (int round = 0; round < max_rounds; ++round) {
        // Blocks walk their share of a list and load the next tile while they work on the current one, so a block should own
        // several tiles: the grid caps are per 64 open images (one GrabCut lane) and grow with the batch a call is given.
        // Late rounds (a handful of open images) are pure launch latency, and empty blocks add to it: never more blocks than tiles.
        const size_t scale = std::max<size_t>(1, ((size_t)n_cur + 32) / 64);
        const int per_image = tl.rt_x * tl.rt_y;
        // The first rounds relabel PARTIALLY: only the work-list launches, without the asynchronous launch that follows the
        // front to its fixpoint.  Labels steer the pushes, they do not have to be exact for the result to be: whatever labels
        // a push phase sees, it turns a valid preflow into a valid preflow, and the cut is read off the EXACT relabel that
        // ends the solve.  A pixel the front has not reached keeps "infinity" for this round — its excess waits (deep inside
        // an object most of it is trapped anyway) — so an image is never closed on a partial relabel, and the round that
        // decides "no active pixel left" is always a full one.  Measured: 57.7 -> 52.9 ms per GrabCut stage (batch 256).
        const bool partial = kn.mf_async && round < kn.mf_partial_rounds;
        const int pr_grid = (int)std::min<size_t>(PUSH_GRID * scale, std::max<size_t>(64, (size_t)n_cur * tl.pt_x * tl.pt_y / 2));
        const int rl_grid = (int)std::min<size_t>(RELAX_GRID * scale, std::max<size_t>(16, cdiv((size_t)n_cur * per_image, 4)));
        // ---- global relabel of the open images
        int relax_launches = 0;
        {
            ProfScope prof(ctx, st, "maxflow_relabel");
            hipLaunchKernelGGL(k_mf_rinit, dim3(cdiv(per_image, 4), n_cur), dim3(256), 0, st, d, tl, list_cur, snk, dist, rl_list[0], rl_flag[0],
                               rl_flag[1], rl_cnt, active, n_open);
            auto relax = [&](int phase) {
                int32_t *li = rl_list[phase & 1], *lo = rl_list[(phase + 1) & 1], *fi = rl_flag[phase & 1], *fo = rl_flag[(phase + 1) & 1];
                if (prof_dev) hipLaunchKernelGGL(k_mf_relax_wave<true>, dim3(rl_grid), dim3(256), 0, st, d, tl, phase, prof_dev, rmask, dirty, rc, dist, rl_cnt, li, lo, fi, fo);
                else hipLaunchKernelGGL(k_mf_relax_wave<false>, dim3(rl_grid), dim3(256), 0, st, d, tl, phase, prof_dev, rmask, dirty, rc, dist, rl_cnt, li, lo, fi, fo);
            };
            // k_mf_rinit starts the labels from the sink links and lists the tiles that have a pixel away from the sink; the
            // first launches relax every listed tile (bandwidth work, plain stores).  With the asynchronous driver the long
            // sparse rest of the front runs inside ONE launch that ends at the exact fixpoint (nothing to read back); without
            // it the host reads the size of the next list every fourth launch.
            int phase = 0;
            if (kn.mf_async && partial) {
                for (; phase < kn.mf_relax_dense; ++phase) relax(phase);
                relax_launches = phase;
            } else if (kn.mf_async) {
                for (; phase < kn.mf_relax_dense; ++phase) relax(phase);
                // (pool size: 512 waves per 64 open images; a lane alone runs the same with 128 or 2048 — the launch is a latency
                // chain, not throughput — and four lanes with larger pools lose to each other's waiting waves: 56.6 -> 58.9 -> 62.8 ms)
                const int grid = (int)std::min<size_t>(ASYNC_GRID * scale, std::max<size_t>(16, cdiv((size_t)n_cur * per_image, 16)));
                if ((rcode = maxflow_relax_async(ctx, st, d, tl, rmask, dirty, rc, dist, rl_cnt + phase % 3, rl_list[phase & 1], rl_flag[phase & 1], ring, aq,
                                                 (int)n_rt, grid, err_flag)))
                    return rcode;
                relax_launches = phase + 1;
            } else {
                for (int guard = 0; guard < 100000; ++guard) {
                    for (int rep = 0; rep < 4; ++rep, ++phase) relax(phase);
                    GGC_LAUNCH_CHECK(ctx);
                    if ((rcode = read_i32(ctx, st, rl_cnt + phase % 3, 1, host))) return rcode;   // size of the next frontier
                    if (host[0] == 0) break;
                }
                relax_launches = phase;
            }
        }
        // ---- who still has work?  (active pixel = excess that can still reach the sink)
        // (active[], the open-image count, the push-list counters and the active total were zeroed by this round's k_mf_rinit)
        hipLaunchKernelGGL(k_mf_active, dim3(std::min(cdiv(tl.pt_x * tl.pt_y, 4), 128), n_cur), dim3(256), 0, st, d, tl, list_cur, ex, dist,
                           active, pt_flag[0], pt_flag[1], pt_list[0], pr_cnt);
        hipLaunchKernelGGL(k_done_update, dim3(cdiv(n_cur, 256)), dim3(256), 0, st, n_cur, list_cur, active, list_nxt, n_open, partial ? 1 : 0, rl_cnt);
        GGC_LAUNCH_CHECK(ctx);
        if ((rcode = read_i32(ctx, st, n_open, 8, host))) return rcode;
        const int n_next = host[0], total_active = host[7];
        if (kn.mf_trace) {   // diagnostics: visit-phase clocks of the dense relabel launches, active pixels / open images per round, the stragglers by image
            long long hh[64 * 8], h[5] = {0, 0, 0, 0, 0};
            GGC_HIP(ctx, hipStreamSynchronize(st));
            GGC_HIP(ctx, hipMemcpy(hh, prof_dev + 64 * 8, sizeof hh, hipMemcpyDeviceToHost));
            GGC_HIP(ctx, hipMemsetAsync(prof_dev + 64 * 8, 0, sizeof hh, st));
            for (int i = 0; i < 64; ++i) for (int k = 0; k < 5; ++k) h[k] += hh[i * 8 + k];
            if (h[3] > 0)
                std::fprintf(stderr, "    [relabel visits] %lld dense visits: per visit load+fill %.2f us, sweeps %.2f us (%.1f sweeps), write-back %.2f us\n",
                             h[3], 0.01 * h[0] / h[3], 0.01 * h[1] / h[3], (double)h[4] / h[3], 0.01 * h[2] / h[3]);
            std::vector<int32_t> act;
            if ((rcode = read_i32(ctx, st, active, B, act))) return rcode;
            const auto t_now = std::chrono::steady_clock::now();
            const double ms = std::chrono::duration<double, std::milli>(t_now - t_prev).count();
            std::fprintf(stderr, "[ggc maxflow] round %d: open images %d, active pixels %d, relabel launches %d, relabel+scan %.3f ms, previous push %.3f ms\n",
                         round, n_next, total_active, relax_launches, ms - push_ms, push_ms);
            t_prev = t_now;
            if (n_next > 0 && n_next <= 12) {
                std::fprintf(stderr, "    [open]");
                for (int b = 0; b < B; ++b) if (act[b]) std::fprintf(stderr, " %d:%d", b, act[b]);
                std::fprintf(stderr, "\n");
            }
        }
        if (n_next == 0) return GGC_OK;
        std::swap(list_cur, list_nxt);
        n_cur = n_next;
        // ---- push-relabel sweeps
        {
            ProfScope prof(ctx, st, "maxflow_push");
            if (kn.mf_async && round > 0 && total_active <= kn.mf_async_push_active) {
                // sparse round: one asynchronous launch chases the excess from tile to tile (chains of at most async_hops hops)
                const int waves = (int)std::min<long long>(4ll * ASYNC_GRID * (long long)scale, std::max<long long>(64, total_active / 4));
                if ((rcode = maxflow_push_async(ctx, st, d, tl, kn.mf_async_tile, kn.mf_async_sweeps, kn.mf_async_hops, rc, ex, snk, dist, rmask, dirty, pr_cnt, pt_list[0], (int)n_pt,
                                                busy, ring, aq, waves, err_flag, prof_dev)))
                    return rcode;
                if (kn.mf_trace) {
                    GGC_HIP(ctx, hipStreamSynchronize(st));
                    push_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prev).count();
                    long long hh[128 * 8], h[7] = {0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
                    GGC_HIP(ctx, hipMemcpy(hh, prof_dev, sizeof hh, hipMemcpyDeviceToHost));
                    GGC_HIP(ctx, hipMemsetAsync(prof_dev, 0, sizeof hh, st));
                    for (int i = 0; i < 64; ++i) for (int k = 0; k < 7; ++k) h[k] += hh[i * 8 + k];
                    for (int i = 0; i < 64; ++i) for (int k = 0; k < 3; ++k) g[k] += hh[(64 + i) * 8 + k];
                    if (h[3] > 0)
                        std::fprintf(stderr, "    [async visit] load+fill %.2f us, sweeps %.2f us (%.1f sweeps), write-back+drain %.2f us\n", 0.01 * g[0] / h[3],
                                     0.01 * g[1] / h[3], (double)g[2] / h[3], 0.01 * (h[1] - g[0] - g[1]) / h[3]);
                    if (h[6] > 0)
                        std::fprintf(stderr, "    [async push] %d waves alive %.1f us on average; %lld visits (%lld followed): per visit wait+lock %.2f us, "
                                     "visit %.2f us, hand-over %.2f us\n", waves, 0.01 * h[5] / h[6], h[3], h[4], h[3] ? 0.01 * h[0] / h[3] : 0.0,
                                     h[3] ? 0.01 * h[1] / h[3] : 0.0, h[3] ? 0.01 * h[2] / h[3] : 0.0);
                }
                continue;
            }
            // dense round.  First round: labels go stale fastest while most excess is still moving, an early relabel pays
            // (8 launches vs 12: +3 %); sweeps per visit measured flat from 6 to 12 and worse either side.  Without the
            // asynchronous driver the tail rounds (few active pixels, labels stay exact) run longer chains of cheap launches.
            const bool tail = !kn.mf_async && total_active <= 4000;
            const int launches = tail ? 64 : (round == 0 ? kn.mf_dense_launches0 : kn.mf_dense_launches);
            // an active pixel opens at most its own tile: empty blocks only add dispatch time to launches that are pure latency
            const int grid = (int)std::min<long long>(pr_grid, std::max<long long>(128, 2ll * total_active));
            for (int phase = 0; phase < launches; ++phase)
                hipLaunchKernelGGL((k_mf_pr_list<1>), dim3(grid), dim3(PT_N), 0, st, d, tl, phase, kn.mf_dense_sweeps, rc, ex, snk, dist, rmask, dirty, pr_cnt,
                                   pt_list[phase & 1], pt_list[(phase + 1) & 1], pt_flag[phase & 1], pt_flag[(phase + 1) & 1]);
            GGC_LAUNCH_CHECK(ctx);
            if (kn.mf_trace) {
                GGC_HIP(ctx, hipStreamSynchronize(st));
                push_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prev).count();
            }
        }
    }
    return set_err(ctx, GGC_E_DEVICE, "max-flow did not converge in %d rounds", max_rounds);
}

} // namespace ggc
