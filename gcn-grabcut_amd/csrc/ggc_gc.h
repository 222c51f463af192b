// ggc_gc.h — shared between ggc_grabcut.hip (GMMs, graph construction) and
// ggc_maxflow.hip (integer push-relabel on the implicit 8-neighbour grid).
#pragma once
#include "ggc_internal.h"

namespace ggc {

constexpr int DINF = 1 << 29;      // "cannot reach the sink"

struct GcDims { int B, H, W, P, n_chunks; };

// directions: 0 left, 1 right, 2 up, 3 down, 4 up-left, 5 down-right, 6 up-right, 7 down-left; rev(dir) = dir ^ 1
__device__ __forceinline__ int dir_dx(int dir) { return (dir == 0 || dir == 4 || dir == 7) ? -1 : ((dir == 1 || dir == 5 || dir == 6) ? 1 : 0); }
__device__ __forceinline__ int dir_dy(int dir) { return (dir == 2 || dir == 4 || dir == 6) ? -1 : ((dir == 3 || dir == 5 || dir == 7) ? 1 : 0); }
__device__ __forceinline__ int dir_nb(const GcDims& d, int y, int x, int dir) {
    const int yy = y + dir_dy(dir), xx = x + dir_dx(dir);
    if (xx < 0 || xx >= d.W || yy < 0 || yy >= d.H) return -1;
    return yy * d.W + xx;
}

// Residual capacities: [B * P][8] — the 8 arcs of a pixel are contiguous (one 32-byte sector).  Every consumer reads all 8
// of a pixel (push visits, arc masks), and the relabel's column reads (arcs that leave a push tile sideways, one pixel per
// image row) then touch one line per row instead of three; as 8 planes [8][B * P] those column reads were 6 us of a 22 us
// relabel visit.
__host__ __device__ __forceinline__ size_t rc_idx(int dir, size_t i) { return i * 8 + (size_t)dir; }

// Maximum preflow + canonical labels for every image with state[b] == 0.
//   rc   [B][P][8] residual capacities (in/out), rc_idx(dir, pixel)     ex, snk [B][P] excess / residual sink capacity (in/out)
//   dist [B][P] out: distance to the sink in the final residual graph, >= DINF when unreachable (=> foreground)
//   rmask [B][P] arc masks of rc (bit dir = residual arc towards dir); masks_exact: every byte is current (cold start) —
//   otherwise the tiles marked dirty by the previous solve keep their mark.  lists [2B], flags [2B+16] scratch;
//   err_flag: device word that receives a non-zero code if an asynchronous launch gives up.
int maxflow(ggc_ctx* ctx, hipStream_t st, const GcDims& d, const int32_t* state, int32_t* rc, int32_t* ex,
            int32_t* snk, int32_t* dist, uint8_t* rmask, int32_t* lists, int32_t* flags, int32_t* err_flag, bool masks_exact);

// Tile geometry of the max-flow kernels (ggc_maxflow.hip: launches over work lists; ggc_maxflow_async.hip: one launch per sparse phase)
constexpr int MF_RT = 32;                          // relabel tile side
constexpr int MF_PT_W = 32, MF_PT_H = 8;           // push tile
struct MfTiles { int rt_x, rt_y, pt_x, pt_y; };    // tiles per image

// up to three int32 regions zeroed by one launch on `st` (ggc_maxflow.hip)
void mf_zero3(hipStream_t st, int32_t* a, size_t na, int32_t* b, size_t nb, int32_t* c, size_t nc);

// Sparse phases as one asynchronous launch each (ggc_maxflow_async.hip): a pool of waves over a ticket queue of tiles.
// Queue control words (int32, every hot counter on its own 128-byte line):
constexpr int AQ_HEAD = 0, AQ_TAIL = 32, AQ_PENDING = 64, AQ_DONE = 96, AQ_BUDGET = 97, AQ_VISITS = 128, AQ_WORDS = 160;
// queue <- list[0 .. *count) (device-side count; the tiles' membership flags in `flag` are set), then ONE launch that ends
// at the relabel's fixpoint.  ring: cap 64-bit slots, cap = number of relabel tiles of the batch.
int maxflow_relax_async(ggc_ctx* ctx, hipStream_t st, const GcDims& d, const MfTiles& tl, uint8_t* rmask, int32_t* dirty, const int32_t* rc, int32_t* dist,
                        const int32_t* count, const int32_t* list, int32_t* flag, unsigned long long* ring, int32_t* q, int cap,
                        int grid, int32_t* err_flag);
// the same for one push phase on 32 x th tiles (th = 16 | 32): chains of at most gen_max tile hops, at most `inner` sweeps per
// visit.  list: the active scan's 32x8 push tiles (n_list_max = their number in the batch); state: one word per tile.
int maxflow_push_async(ggc_ctx* ctx, hipStream_t st, const GcDims& d, const MfTiles& tl, int th, int inner, int gen_max, int32_t* rc,
                       int32_t* ex, int32_t* snk, int32_t* dist, uint8_t* rmask, int32_t* dirty, const int32_t* count, const int32_t* list, int n_list_max, int32_t* state,
                       unsigned long long* ring, int32_t* q, int waves, int32_t* err_flag, long long* prof = nullptr);

} // namespace ggc
