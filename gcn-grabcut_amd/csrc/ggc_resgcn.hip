// ggc_resgcn.hip — ResGCNNet forward (reference model.py:421-546, eval mode) as
// hand-written gfx950 kernels.
//
// Kernel map:
//   k_input        M1  BatchNorm-eval -> Linear 19->D -> LayerNorm -> GELU -> prior booster
//   k_edge_gate    M2  EdgeContext: edge MLP, mean over incoming edges, LN, Linear, sigmoid
//   k_gemm<.,0>    M3  LayerNorm(h) @ W^T on f32 MFMA (v_mfma_f32_32x32x2_f32)
//   k_aggregate<.,0>  M3  GCNConv scatter-gather over the destination CSR with the
//                      fused epilogue h + gelu((agg + b) * gate)       <- graded kernel
//   k_aggregate<.,1>  M4  SAGEConv mean aggregation
//   k_gemm<.,1>    M4  lin_l(mean) + lin_r(h), LayerNorm, GELU (two MFMA phases)
//   k_jk           M5  softmax(jk_logits)-weighted sum of the n+2 states + attention score
//   k_graph_ctx    M6  per-graph softmax readout, compress/expand MLP
//   k_gemm<.,2>    M7  (h_jk * g) -> LN -> Linear -> GELU -> head -> softmax
//
// Data layout in HBM: node-major row vectors, f32, D contiguous (512 B rows at
// D=128).  Graphs of a batch are concatenated (PyG Batch semantics); node_ptr
// gives the per-graph ranges.  The destination CSR (row_ptr, col, eid) is
// built once per forward, stable in edge order, and reused by all 8 gathers.
#include "ggc_internal.h"
#include <atomic>
#include <cmath>
#include <type_traits>

namespace ggc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IN_CH = 19, EDGE_CH = 5, N_PRIOR = 3, N_CLS = 3;

// ------------------------------------------------------------------ CSR build

__global__ void k_count_dst(int E, const int32_t* __restrict__ dst, int32_t* __restrict__ cnt) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x)
        atomicAdd(&cnt[dst[e]], 1);
}

// Exclusive scan of cnt[0..n) into out[0..n], out[n] = total, in three small launches: per-block totals of SCAN_BLOCK
// elements, a one-block scan of the totals, and the scan of each block from its base.  (One block over all 154 k rows of a
// batch-256 forward took 233 us; this is a few microseconds per launch.)  part: cdiv(n, SCAN_BLOCK) + 1 words.
constexpr int SCAN_BLOCK = 2048;       // 256 threads x 8 elements
__global__ void __launch_bounds__(256) k_scan_totals(int n, const int32_t* __restrict__ cnt, int32_t* __restrict__ part) {
    __shared__ int32_t red[4];
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 8;
    int32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (base + j < n) s += cnt[base + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(1024) k_scan_parts(int m, int32_t* __restrict__ part) {      // exclusive, in place; part[m] = total
    __shared__ int32_t sh[1024];
    const int tid = threadIdx.x;
    const int chunk = (m + 1023) / 1024, beg = tid * chunk, end = min(beg + chunk, m);
    int32_t s = 0;
    for (int i = beg; i < end; ++i) s += part[i];
    sh[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int32_t v = (tid >= off) ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    int32_t run = (tid == 0) ? 0 : sh[tid - 1];
    for (int i = beg; i < end; ++i) { const int32_t c = part[i]; part[i] = run; run += c; }
    if (tid == 1023) part[m] = sh[1023];
}
__global__ void __launch_bounds__(256) k_scan_blocks(int n, const int32_t* __restrict__ cnt, const int32_t* __restrict__ part,
                                                     int m, int32_t* __restrict__ out) {
    __shared__ int32_t wsum[4];
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 8, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t c[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { c[j] = base + j < n ? cnt[base + j] : 0; s += c[j]; }
    int32_t incl = s;                                      // inclusive scan of the threads' sums inside the wave
    for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t run = part[blockIdx.x] + (incl - s);
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (base + j < n) out[base + j] = run; run += c[j]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = part[m];
}
static void exclusive_scan(hipStream_t st, int n, const int32_t* cnt, int32_t* part, int32_t* out) {
    const int m = cdiv(std::max(n, 1), SCAN_BLOCK);
    hipLaunchKernelGGL(k_scan_totals, dim3(m), dim3(256), 0, st, n, cnt, part);
    hipLaunchKernelGGL(k_scan_parts, dim3(1), dim3(1024), 0, st, m, part);
    hipLaunchKernelGGL(k_scan_blocks, dim3(m), dim3(256), 0, st, n, cnt, part, m, out);
}

__global__ void k_fill_eid(int E, const int32_t* __restrict__ dst, const int32_t* __restrict__ row_ptr,
                           int32_t* __restrict__ cursor, int32_t* __restrict__ eid) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int d = dst[e];
        const int pos = row_ptr[d] + atomicAdd(&cursor[d], 1);
        eid[pos] = e;
    }
}

// One thread per row: restore edge order inside the row (stable CSR), emit col and dis.
__global__ void k_sort_rows(int N, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ eid,
                            const int32_t* __restrict__ src, int32_t* __restrict__ col,
                            float* __restrict__ dis) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < N; r += gridDim.x * blockDim.x) {
        const int beg = row_ptr[r], end = row_ptr[r + 1];
        for (int i = beg + 1; i < end; ++i) {
            const int v = eid[i];
            int j = i - 1;
            while (j >= beg && eid[j] > v) { eid[j + 1] = eid[j]; --j; }
            eid[j + 1] = v;
        }
        for (int i = beg; i < end; ++i) col[i] = src[eid[i]];
        if (dis) dis[r] = 1.0f / sqrtf((float)(end - beg) + 1.0f);
    }
}

__global__ void k_fill_batch(int G, int N, const int32_t* __restrict__ node_ptr, int32_t* __restrict__ batch) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        int lo = 0, hi = G; // find g with node_ptr[g] <= i < node_ptr[g+1]
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (node_ptr[mid] <= i) lo = mid; else hi = mid; }
        batch[i] = lo;
    }
}

static int build_csr(ggc_ctx* ctx, hipStream_t st, int N, int E, const int32_t* src, const int32_t* dst,
                     int32_t* row_ptr, int32_t* col, int32_t* eid, int32_t* cursor, float* dis) {
    GGC_HIP(ctx, hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)(N + 1), st));
    if (E > 0) {
        hipLaunchKernelGGL(k_count_dst, dim3(min(cdiv(E, 256), 4096)), dim3(256), 0, st, E, dst, cursor);
        GGC_LAUNCH_CHECK(ctx);
    }
    int32_t* scan_part = scratch_t<int32_t>(ctx, S_CSR_SCAN, (size_t)cdiv(std::max(N, 1), SCAN_BLOCK) + 2);
    if (!scan_part) return GGC_E_OOM;
    exclusive_scan(st, N, cursor, scan_part, row_ptr);
    GGC_LAUNCH_CHECK(ctx);
    GGC_HIP(ctx, hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)(N + 1), st));
    if (E > 0) {
        hipLaunchKernelGGL(k_fill_eid, dim3(min(cdiv(E, 256), 4096)), dim3(256), 0, st, E, dst, row_ptr, cursor, eid);
        GGC_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(k_sort_rows, dim3(min(cdiv(N, 256), 4096)), dim3(256), 0, st, N, row_ptr, eid, src, col, dis);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

// --------------------------------------------------------------- M1: input stage
// One wave per node; lane owns columns lane + 64 j.
struct InputW {
    const float *bn_w, *bn_b, *bn_rm, *bn_rv;
    const float *w_inT /*[19][D]*/, *b_in, *ln_w, *ln_b;
    const float *pb_w0 /*[Q,3]*/, *pb_b0, *pb_w2T /*[Q][D]*/, *pb_b2;
};

// Dt: the model's true width.  D is Dt rounded up to a multiple of 32 (MFMA tiling); the padded channels carry exact zeros
// through every layer (zero weight rows / columns, zero LayerNorm weights), so only the LayerNorm STATISTICS need to know
// Dt: the sum over D equals the sum over Dt, the mean divides by Dt, and the squared deviations of the padded channels
// (each mean^2) are left out.
template <int D>
__global__ void __launch_bounds__(256) k_input(int N, const float* __restrict__ x, InputW w, int Q, int Dt,
                                               float* __restrict__ h) {
    constexpr int NC = (D + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int node = wave; node < N; node += n_waves) {
        const float* xi = x + (size_t)node * IN_CH;
        float xn[IN_CH];
#pragma unroll
        for (int k = 0; k < IN_CH; ++k)
            xn[k] = (xi[k] - w.bn_rm[k]) / sqrtf(w.bn_rv[k] + 1e-5f) * w.bn_w[k] + w.bn_b[k];
        float a[NC];
        float s1 = 0.0f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            float acc = 0.0f;
            if (c < D) {
#pragma unroll
                for (int k = 0; k < IN_CH; ++k) acc += xn[k] * w.w_inT[k * D + c];
                acc += w.b_in[c];
                s1 += acc;
            }
            a[j] = acc;
        }
        const float mean = wave_sum(s1) / (float)Dt;
        float s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            if (c < Dt) { const float d = a[j] - mean; s2 += d * d; }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)Dt + 1e-5f);
        // prior booster hidden unit on lane q < Q
        const float p0 = xi[IN_CH - 3], p1 = xi[IN_CH - 2], p2 = xi[IN_CH - 1];
        float bq = 0.0f;
        if (lane < Q) {
            float acc = 0.0f;
            acc += p0 * w.pb_w0[lane * 3 + 0];
            acc += p1 * w.pb_w0[lane * 3 + 1];
            acc += p2 * w.pb_w0[lane * 3 + 2];
            bq = gelu_f(acc + w.pb_b0[lane]);
        }
        float g[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) g[j] = 0.0f;
        for (int q = 0; q < Q; ++q) {
            const float bv = __shfl(bq, q, 64);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = lane + 64 * j;
                if (c < D) g[j] += bv * w.pb_w2T[q * D + c];
            }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                const float v = gelu_f((a[j] - mean) * rstd * w.ln_w[c] + w.ln_b[c]);
                h[(size_t)node * D + c] = v * (1.0f + sigmoid_f(g[j] + w.pb_b2[c]));
            }
        }
    }
}

// ------------------------------------------------------------- M2: edge context
struct EdgeW {
    const float *w0 /*[C,5]*/, *b0, *w2T /*[C][C] in-major*/, *b2;
    const float *ln_w, *ln_b, *wgT /*[C][D]*/, *bg;
};

// A wave owns EC_NODES consecutive nodes at a time, lane = context channel (C <= 64).  Both weight matrices are staged once
// per block in LDS and every k-step of the two matrix-vector products serves all EC_NODES nodes: one LDS read of the weight,
// a v_readlane broadcast of each node's value, and the same multiply-add sequence per output as the one-node-per-wave
// form (products added in k order from zero, bias last) — so the gate keeps its bits.  That form re-read the 48 KB of
// weights from the cache for every node: 0.76 ms per batch-256 forward.
constexpr int EC_NODES = 4;
template <int D>
__global__ void __launch_bounds__(256) k_edge_gate(int N, const int32_t* __restrict__ row_ptr,
                                                   const int32_t* __restrict__ eid,
                                                   const float* __restrict__ edge_attr, EdgeW w, int C,
                                                   float* __restrict__ gate) {
    constexpr int NC = (D + 63) / 64;
    extern __shared__ float s_w[];                  // w2T [C][C] | wgT [C][D]
    float* s_w2 = s_w;
    float* s_wg = s_w + C * C;
    for (int i = threadIdx.x; i < C * C; i += 256) s_w2[i] = w.w2T[i];
    for (int i = threadIdx.x; i < C * D; i += 256) s_wg[i] = w.wgT[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    float w0[EDGE_CH] = {0, 0, 0, 0, 0};
    float b0 = 0.0f, b2 = 0.0f, lw = 0.0f, lb = 0.0f;
    if (lane < C) {
#pragma unroll
        for (int k = 0; k < EDGE_CH; ++k) w0[k] = w.w0[lane * EDGE_CH + k];
        b0 = w.b0[lane]; b2 = w.b2[lane]; lw = w.ln_w[lane]; lb = w.ln_b[lane];
    }
    float bg[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) bg[j] = lane + 64 * j < D ? w.bg[lane + 64 * j] : 0.0f;
    for (int n0 = wave * EC_NODES; n0 < N; n0 += n_waves * EC_NODES) {
        float m[EC_NODES];
        int cnt[EC_NODES];
        // the nodes' incoming edges are one contiguous stretch of the CSR: a lane fetches one edge (index + 5 attributes, the
        // loads of a whole stretch in flight together), the per-edge loop then takes them from registers by v_readlane —
        // walking the edges one at a time through scalar loads was two dependent memory round trips per edge
        int rp[EC_NODES + 1];
#pragma unroll
        for (int r = 0; r <= EC_NODES; ++r) rp[r] = __builtin_amdgcn_readfirstlane(row_ptr[min(n0 + r, N)]);
        float sum[EC_NODES];
#pragma unroll
        for (int r = 0; r < EC_NODES; ++r) sum[r] = 0.0f;
        for (int chunk = rp[0]; chunk < rp[EC_NODES]; chunk += 64) {
            const int p = chunk + lane;
            float a[EDGE_CH];
            {
                const bool valid = p < rp[EC_NODES];
                const size_t e = valid ? (size_t)eid[p] : 0;
#pragma unroll
                for (int k = 0; k < EDGE_CH; ++k) a[k] = valid ? edge_attr[e * EDGE_CH + k] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < EC_NODES; ++r) {
                const int q0 = max(rp[r], chunk) - chunk, q1 = min(rp[r + 1], chunk + 64) - chunk;      // this node's edges inside the chunk
                for (int q = q0; q < q1; ++q) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < EDGE_CH; ++k) acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[k]), q)) * w0[k];
                    sum[r] += gelu_f(acc + b0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < EC_NODES; ++r) {
            cnt[r] = rp[r + 1] - rp[r];
            // mean over incoming edges commutes with the second (linear) layer:
            // mean_e(W2 e1_e + b2) = W2 mean_e(e1_e) + b2  (model.py:138, _scatter_mean :69-74)
            m[r] = sum[r] / (float)(cnt[r] > 0 ? cnt[r] : 1);
        }
        float acc[EC_NODES];
#pragma unroll
        for (int r = 0; r < EC_NODES; ++r) acc[r] = 0.0f;
        for (int k = 0; k < C; ++k) {
            const float wk = lane < C ? s_w2[k * C + lane] : 0.0f;
#pragma unroll
            for (int r = 0; r < EC_NODES; ++r)
                acc[r] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m[r]), k)) * wk;
        }
        float ln[EC_NODES];
#pragma unroll
        for (int r = 0; r < EC_NODES; ++r) {
            const float cv = (lane < C && cnt[r] > 0) ? acc[r] + b2 : 0.0f;
            const float mean = wave_sum(lane < C ? cv : 0.0f) / (float)C;
            const float dv = (lane < C) ? cv - mean : 0.0f;
            const float rstd = 1.0f / sqrtf(wave_sum(dv * dv) / (float)C + 1e-5f);
            ln[r] = (lane < C) ? dv * rstd * lw + lb : 0.0f;
        }
        float g[EC_NODES][NC];
#pragma unroll
        for (int r = 0; r < EC_NODES; ++r)
#pragma unroll
            for (int j = 0; j < NC; ++j) g[r][j] = 0.0f;
        for (int k = 0; k < C; ++k) {
            float wk[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) wk[j] = lane + 64 * j < D ? s_wg[k * D + lane + 64 * j] : 0.0f;
#pragma unroll
            for (int r = 0; r < EC_NODES; ++r) {
                const float lk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ln[r]), k));
#pragma unroll
                for (int j = 0; j < NC; ++j) g[r][j] += lk * wk[j];
            }
        }
#pragma unroll
        for (int r = 0; r < EC_NODES; ++r) {
            if (n0 + r >= N) break;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = lane + 64 * j;
                if (c < D) gate[(size_t)(n0 + r) * D + c] = sigmoid_f(g[r][j] + bg[j]);
            }
        }
    }
}

// ---------------------------------------------------------- M3/M4/M7: MFMA GEMM
// out[N,D] = op(A)[N,D] @ W^T, f32 in / f32 accumulate on v_mfma_f32_32x32x2_f32.
// A block is 4 waves; each wave owns 32 rows x D columns (T = D/32 accumulator
// tiles).  The k index is permuted so that lane half hk = lane>>5 covers
// k in [hk*D/2, (hk+1)*D/2): each lane then reads one contiguous half-row of A
// (float4 loads) and W is pre-packed on the host as Wp[s/4][t][lane][s%4] =
// W[32t + (lane&31)][hk*D/2 + s], staged once per block into LDS and read with
// conflict-free ds_read_b128.
//   MODE 0: A = LayerNorm(A1); store                                   (GCN XW)
//   MODE 1: A1 @ W1^T + A2 @ W2^T + bias -> LayerNorm -> GELU          (SAGE)
//   MODE 2: A = LayerNorm(A1 * gvec[batch]); + bias -> GELU -> head -> softmax
//   MODE 4: as MODE 2 without the LayerNorm (GATTrimapNet head)
struct GemmArgs {
    const float *A1, *A2, *Wp1, *Wp2;
    const float *ln_w, *ln_b;        // prologue LayerNorm
    const float *bias;               // [D]
    const int32_t* batch;            // MODE 2
    const float* gvec;               // MODE 2: [G,D]
    const float *ep_w, *ep_b;        // MODE 1: LN weight/bias; MODE 2: head weight [3,D] / bias [3]
    float *out, *out2;               // MODE 2: logits / probs (either may be null)
    int accumulate = 0;              // MODE 3: out += A W^T instead of out = A W^T
    int Dt = 0;                      // true width for the LayerNorm statistics (0: D), see k_input
};

#ifdef GEMM_TRACE     // experiment build: per-wave phase stamps of k_gemm<128, 0> kept in SGPRs (no scheduling fences)
__device__ unsigned long long g_gemm_trace[8192 * 8];
#define GEMM_STAMP(i) do { if (D == 128 && MODE == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i]) :: "memory"); } while (0)
#else
#define GEMM_STAMP(i) do {} while (0)
#endif
template <int D, int MODE>
__global__ void __launch_bounds__(256) k_gemm(int N, GemmArgs g) {
#ifdef GEMM_TRACE
    unsigned long long stamp[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr int T = D / 32, KH = D / 2;
    extern __shared__ float4 smem4[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    const int row = min(r0 + li, N - 1);
    GEMM_STAMP(0);

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    constexpr int PHASES = (MODE == 1) ? 2 : 1;
#pragma unroll
    for (int ph = 0; ph < PHASES; ++ph) {
        const float* Wp = ph ? g.Wp2 : g.Wp1;
        const float* A = ph ? g.A2 : g.A1;
        // the wave's rows are requested first: their latency runs under the staging of W
        float4 av[KH / 4];
        {
            const float4* ap = reinterpret_cast<const float4*>(A + (size_t)row * D + hk * KH);
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) av[q] = ap[q];
        }
        if (ph) __syncthreads();
        {   // all D * D / 1024 loads of a thread in flight at once (rolled, the loop paid one memory round trip per 4 KB: 17 k cycles at D = 128)
            float4 wv[D * D / 1024];
#pragma unroll
            for (int q = 0; q < D * D / 1024; ++q) wv[q] = reinterpret_cast<const float4*>(Wp)[tid + q * 256];
#pragma unroll
            for (int q = 0; q < D * D / 1024; ++q) smem4[tid + q * 256] = wv[q];
        }
        __syncthreads();
        GEMM_STAMP(1);

        float a[KH];
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            const float4 v = av[q];
            a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
        }
        if (MODE == 2 || MODE == 4) {
            const float* gv = g.gvec + (size_t)g.batch[row] * D + hk * KH;
#pragma unroll
            for (int s = 0; s < KH; ++s) a[s] *= gv[s];
        }
        if (MODE == 0 || MODE == 2) {
            float s1 = 0.0f;
#pragma unroll
            for (int s = 0; s < KH; ++s) s1 += a[s];
            s1 += __shfl_xor(s1, 32, 64);
#ifdef GEMM_TRACE
            if (D == 128 && MODE == 0) asm volatile("s_nop 0" :: "v"(s1));       // the stamp below waits for the row sum, i.e. for the loads
#endif
            GEMM_STAMP(2);
            const int dt = g.Dt > 0 ? g.Dt : D, nv = min(max(dt - hk * KH, 0), KH);   // this lane's channels below the true width
            const float mean = s1 / (float)dt;
            float s2 = 0.0f;
#pragma unroll
            for (int s = 0; s < KH; ++s) { const float d = a[s] - mean; s2 += (s < nv) ? d * d : 0.0f; }
            s2 += __shfl_xor(s2, 32, 64);
            const float rstd = 1.0f / sqrtf(s2 / (float)dt + 1e-5f);
            const float* lw = g.ln_w + hk * KH;
            const float* lb = g.ln_b + hk * KH;
#pragma unroll
            for (int s = 0; s < KH; ++s) a[s] = (a[s] - mean) * rstd * lw[s] + lb[s];
        }
#ifdef GEMM_TRACE
        if (D == 128 && MODE == 0) asm volatile("s_nop 0" :: "v"(a[0]), "v"(a[KH - 1]));
#endif
        GEMM_STAMP(3);
        // Consecutive MFMAs go to different accumulators (the T column tiles of one k step); each accumulator still sees its
        // k steps in the same order.  (Phase stamps, GEMM_TRACE: the MFMA loops of the two waves of a SIMD never overlap, in this
        // order or with the four k steps of a B fragment back to back on one accumulator, and the other wave's LayerNorm
        // crawls meanwhile — the f32 MFMA runs at the vector rate and leaves the SIMD's vector issue little room.)
#pragma unroll
        for (int s4 = 0; s4 < KH / 4; ++s4) {
            float bq[T][4];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float4 b = smem4[(s4 * T + t) * 64 + lane];
                bq[t][0] = b.x; bq[t][1] = b.y; bq[t][2] = b.z; bq[t][3] = b.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < T; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4 + j], bq[t][j], acc[t], 0, 0, 0);
        }
    }

#ifdef GEMM_TRACE
    if (D == 128 && MODE == 0) asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[T - 1][15]));
#endif
    GEMM_STAMP(4);
    // C/D layout: col = 32 t + (lane & 31), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (MODE == 3) {                 // plain product (GCNTrimapNet: no prologue norm), optionally accumulated
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int grow = r0 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            if (grow < N) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    float* o = g.out + (size_t)grow * D + 32 * t + li;
                    *o = g.accumulate ? *o + acc[t][r] : acc[t][r];
                }
            }
        }
    } else if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int grow = r0 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            if (grow < N) {
#pragma unroll
                for (int t = 0; t < T; ++t) g.out[(size_t)grow * D + 32 * t + li] = acc[t][r];
            }
        }
    } else if (MODE == 1) {
        float bias[T], lw[T], lb[T];
#pragma unroll
        for (int t = 0; t < T; ++t) { bias[t] = g.bias[32 * t + li]; lw[t] = g.ep_w[32 * t + li]; lb[t] = g.ep_b[32 * t + li]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s1 = 0.0f;
#pragma unroll
            for (int t = 0; t < T; ++t) { acc[t][r] += bias[t]; s1 += acc[t][r]; }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
            const int dt = g.Dt > 0 ? g.Dt : D;
            const float mean = s1 / (float)dt;
            float s2 = 0.0f;
#pragma unroll
            for (int t = 0; t < T; ++t) { const float d = acc[t][r] - mean; s2 += (32 * t + li < dt) ? d * d : 0.0f; }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
            const float rstd = 1.0f / sqrtf(s2 / (float)dt + 1e-5f);
            const int grow = r0 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            if (grow < N) {
#pragma unroll
                for (int t = 0; t < T; ++t)
                    g.out[(size_t)grow * D + 32 * t + li] = gelu_f((acc[t][r] - mean) * rstd * lw[t] + lb[t]);
            }
        }
    } else {
        float bias[T], hw[N_CLS][T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            bias[t] = g.bias[32 * t + li];
#pragma unroll
            for (int c = 0; c < N_CLS; ++c) hw[c][t] = g.ep_w[c * D + 32 * t + li];
        }
        const float hb0 = g.ep_b[0], hb1 = g.ep_b[1], hb2 = g.ep_b[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p[N_CLS] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float v = gelu_f(acc[t][r] + bias[t]);
#pragma unroll
                for (int c = 0; c < N_CLS; ++c) p[c] += v * hw[c][t];
            }
#pragma unroll
            for (int c = 0; c < N_CLS; ++c)
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) p[c] += __shfl_xor(p[c], o, 64);
            const int grow = r0 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            if (li == 0 && grow < N) {
                const float l0 = p[0] + hb0, l1 = p[1] + hb1, l2 = p[2] + hb2;
                if (g.out) { g.out[(size_t)grow * 3 + 0] = l0; g.out[(size_t)grow * 3 + 1] = l1; g.out[(size_t)grow * 3 + 2] = l2; }
                if (g.out2) {
                    const float mx = fmaxf(l0, fmaxf(l1, l2));
                    const float e0 = ggc_expf(l0 - mx), e1 = ggc_expf(l1 - mx), e2 = ggc_expf(l2 - mx);
                    const float s = (e0 + e1) + e2;
                    g.out2[(size_t)grow * 3 + 0] = e0 / s; g.out2[(size_t)grow * 3 + 1] = e1 / s; g.out2[(size_t)grow * 3 + 2] = e2 / s;
                }
            }
        }
    }
#ifdef GEMM_TRACE
    GEMM_STAMP(5);
    __builtin_amdgcn_s_waitcnt(0);
    GEMM_STAMP(6);
    if (D == 128 && MODE == 0 && lane == 0 && blockIdx.x * 4 + wave < 8192) {
        unsigned long long* o = g_gemm_trace + (blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 7; ++i) o[i] = stamp[i];
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
    }
#endif
}

// ------------------------------------------------- M3/M4: CSR scatter-gather
// Destination-major gather: LPR lanes x float4 cover one D-wide row, a wave
// handles 64/LPR rows, a block of 4 waves handles RPB consecutive rows.  Blocks
// are remapped so each XCD walks a contiguous range of rows (= whole images):
// the ~11 re-reads of every xw row then hit that XCD's L2.  Neighbours are
// summed in CSR (= edge) order with one rounding per multiply and per add, the
// self loop last, exactly like the oracle.
//   MODE 0 (GCNConv): out = sum dis[j] dis[i] xw[j] + dis[i]^2 xw[i] + bias,
//                     fused epilogue h + gelu(out * gate) when gate != null.
//   MODE 1 (SAGE mean): out = sum xw[j] / max(cnt, 1).
template <int D> struct AggCfg {
    static constexpr int LPR = (D <= 32) ? 8 : (D <= 64) ? 16 : 32;
    static constexpr int RPW = 64 / LPR;
    static constexpr int RPB = 4 * RPW;
};

__device__ __forceinline__ void fma4(float4& acc, float w, const float4& v) {
    acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
}

template <int D, int MODE>
__global__ void __launch_bounds__(1024) k_aggregate(int N, const float* __restrict__ xw,
                                                   const int32_t* __restrict__ row_ptr,
                                                   const int32_t* __restrict__ col,
                                                   const float* __restrict__ dis,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ gate,
                                                   const float* __restrict__ h,
                                                   float* __restrict__ out) {
    constexpr int LPR = AggCfg<D>::LPR, RPW = AggCfg<D>::RPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int row = blk * (RPW * (int)(blockDim.x >> 6)) + wave * RPW + sub;
    // The sub-group of LPR lanes that owns a row first loads up to LPR column indices (and their
    // dis weights) with ONE coalesced load each, then broadcasts them by shuffle: the dependent
    // chain per row is row_ptr -> col -> dis -> rows instead of one col/dis round trip per
    // neighbour, and up to 8 neighbour rows are in flight per lane.  Summation order is unchanged
    // (CSR = edge order, self loop last).
    const bool rvalid = row < N, cvalid = sl * 4 < D;
    const int rowc = rvalid ? row : 0, slc = cvalid ? sl : 0;
    const int beg = rvalid ? row_ptr[rowc] : 0, end = rvalid ? row_ptr[rowc + 1] : 0;
    const float di = (MODE == 0) ? dis[rowc] : 1.0f;
    constexpr int D4 = D / 4;
    const float4* xw4 = reinterpret_cast<const float4*>(xw) + slc;
    const size_t o4 = (size_t)rowc * D4 + slc;
    float4 self = make_float4(0.f, 0.f, 0.f, 0.f), gt = self, hv = self;
    if (MODE == 0) {                     // epilogue operands: issue their loads before the gather
        self = xw4[(size_t)rowc * D4];
        if (gate) { gt = reinterpret_cast<const float4*>(gate)[o4]; hv = reinterpret_cast<const float4*>(h)[o4]; }
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = beg; base < end; base += LPR) {
        const int n = min(LPR, end - base);
        int c = 0;
        float w = 0.0f;
        if (sl < n) { c = col[base + sl]; if (MODE == 0) w = dis[c] * di; }
        for (int k = 0; k < n; k += 8) {
            float4 v[8];
            float wk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = min(k + u, n - 1);
                const int j = __shfl(c, kk, LPR);
                wk[u] = __shfl(w, kk, LPR);
                if (k + u < n) v[u] = xw4[(size_t)j * D4];     // predicated: no duplicate row fetches in the tail
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k + u < n) {
                    if (MODE == 0) fma4(acc, wk[u], v[u]);
                    else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
            }
        }
    }
    if (MODE == 0) {
        fma4(acc, di * di, self);
        if (bias) {
            const float4 b = reinterpret_cast<const float4*>(bias)[slc];
            acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
        }
        if (gate) {
            acc.x = hv.x + gelu_f(acc.x * gt.x); acc.y = hv.y + gelu_f(acc.y * gt.y);
            acc.z = hv.z + gelu_f(acc.z * gt.z); acc.w = hv.w + gelu_f(acc.w * gt.w);
        }
    } else {
        const int cnt = end - beg;
        const float c = (float)(cnt > 0 ? cnt : 1);
        acc.x /= c; acc.y /= c; acc.z /= c; acc.w /= c;
    }
    if (rvalid && cvalid) reinterpret_cast<float4*>(out)[o4] = acc;
}

// Graph-resident form of the gather (the one the forward pass uses).  PMC profiling of the kernel
// above shows ideal HBM traffic (FETCH+WRITE == algorithmic bytes) but the per-CU L1/texture path
// ~76 % busy and two thirds of the L2 requests being re-reads: every neighbour row costs a 16-cycle
// slot of the 64 B/clk vector-memory pipe.  The graphs of a batch are independent and small (about
// 600 superpixels), so a block owns ONE graph and a SW-float column slice of the feature matrix:
// it copies its [n_g, SW] slice of xw (and the graph's dis) into LDS once by LDS-DMA (128-B segments,
// every byte of xw is read from memory exactly once) and then serves every neighbour row and the
// self loop by ds_read_b128 on the 256 B/clk LDS pipe.  Two blocks share a CU (80 KiB each), so one
// block's fill overlaps the other's gather.  The D/SW slices of one graph run on the same XCD and
// share the CSR through its L2.  A graph with more than CAP nodes, or an edge that leaves its graph,
// falls back to global loads per block / per row.  Arithmetic and summation order are those of
// k_aggregate.

// Block shape of the 32-float slices (round 3, bench batch, us per launch of the gated kernel on one box): 512 threads with 8
// neighbour rows per batch of LDS reads and the epilogue operands two rows ahead (116 VGPRs, 16 waves per CU) 60.3-62.0;
// 256 threads 72-75 whatever the prefetch depth (2 / 4 / 6 rows): the kernel is short of WAVES, not of bytes in flight;
// 768 threads, 2 rows per batch (76 VGPRs, 24 waves) 59.3-59.5; 1024 threads, 2 rows per batch, one row ahead (62 VGPRs, the
// full 32 waves per CU) 57.6-58.0.  Anything that spills (768 / 1024 threads with 4 or 8 rows per batch) loses 15 us.
#ifndef AGG_THREADS
#define AGG_THREADS 1024
#endif
#ifndef AGG_PF
#define AGG_PF 1
#endif
#ifndef AGG_UB
#define AGG_UB 2
#endif
// (16-float slices — 64-byte fill segments, three 52-KB blocks per CU, meant for graphs of 620-782 nodes — took 143-150 us on
// the same batch against 96 us for the direct gather k_aggregate: that route is gone, such batches gather directly.)
template <int SW> struct AggGraph {
    static_assert(SW == 32, "one slice width is built");
    static constexpr int T = AGG_THREADS, NW = T / 64;                         // threads per block
    static constexpr int PF = AGG_PF;                                          // epilogue rows in flight ahead of the gather
    static constexpr int UB = AGG_UB;                                          // neighbour rows per batch of LDS reads
    static constexpr int LPR = SW / 4, RPW = 64 / LPR, RPP = NW * RPW;         // rows per pass of the block
    static constexpr int LDS = 80 * 1024;                                      // 2 blocks per CU
    static constexpr int OCC = 2 * T / 256 < 8 ? 2 * T / 256 : 8;              // waves per SIMD the blocks of a CU need
    static constexpr int CAP_LDS = (LDS - SW * 4 - 4) / (SW * 4 + 4);          // tile row + dis entry, one zero row
    static constexpr int CAP = CAP_LDS < 1023 ? CAP_LDS : 1023;                // 10-bit row offsets in the packed columns
    static constexpr int K = (CAP + RPP - 1) / RPP;                            // passes for a full tile
    static constexpr int FILL = (CAP * LPR + T - 1) / T;                       // LDS-DMA pieces per thread
};

// broadcast lane U of every group of W lanes (W = 4 or 8): ds_swizzle, no address VGPR and no VALU
template <int W, int U> __device__ __forceinline__ int group_bcast(int v) {
    return __builtin_amdgcn_ds_swizzle(v, (0x1f & ~(W - 1)) | (U << 5));
}
template <int W, int U> __device__ __forceinline__ float group_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (0x1f & ~(W - 1)) | (U << 5)));
}
template <int W, int U0, int NU, typename F> __device__ __forceinline__ void unroll_bcast(F&& f) {
    if constexpr (NU > 0) { f(std::integral_constant<int, U0>{}); unroll_bcast<W, U0 + 1, NU - 1>(f); }
}

// Per (row, lane) column words of the graph-resident gather, built once per forward pass and shared by
// every layer (7 aggregations read the same CSR): for lane sl of the LPR lanes that own a row,
//   bits 0-9 / 10-19: tile row (offset inside the graph) of neighbours sl and LPR + sl, ZROW when there is none
//   bits 20+        : n + 1, or 0 for an irregular row (more than 2 * LPR neighbours, or an edge that leaves
//                     the graph) which the kernel handles by the generic route
// One coalesced load per row replaces the dependent row_ptr -> col hops in the gather kernel.
template <int SW>
__global__ void __launch_bounds__(256) k_agg_pack(int N, const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ batch,
                                                  const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                  int32_t* __restrict__ pack) {
    constexpr int LPR = AggGraph<SW>::LPR, ZROW = AggGraph<SW>::CAP;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // N * LPR is a multiple of LPR: whole groups stay together
    const int row = min(i / LPR, N - 1), sl = i % LPR;
    const int g = batch[row];
    const int g0 = node_ptr[g], n_g = node_ptr[g + 1] - g0;
    const int beg = row_ptr[row], n = row_ptr[row + 1] - beg;
    bool regular = n <= 2 * LPR;
    int pk = 0;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const bool have = nb * LPR + sl < n;
        const int off = have ? col[beg + nb * LPR + sl] - g0 : 0;
        regular = regular && (unsigned)off < (unsigned)n_g;
        pk |= (have ? off & 1023 : ZROW) << (10 * nb);
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
        const int other = __shfl_xor((int)regular, o, LPR);     // every lane takes part: no short-circuit around the shuffle
        regular = regular && other != 0;
    }
    if (i < N * LPR) pack[i] = regular ? pk | (n + 1) << 20 : (ZROW | ZROW << 10);
}

// One row of the column slice by the generic route (columns, weights and out-of-tile rows from global memory).
template <int D, int MODE, int SW>
__device__ __forceinline__ void agg_row_generic(int row, int g0, int n_g, const float4* tile, const float4* __restrict__ xw4g,
                                                const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                const float* __restrict__ dis, const float4& bias4,
                                                const float* __restrict__ gate, const float* __restrict__ h,
                                                float* __restrict__ out, int s, int sl) {
    constexpr int LPR = SW / 4, D4 = D / 4;
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    const float di = (MODE == 0) ? dis[row] : 1.0f;
    const size_t o4 = (size_t)row * D4 + s * LPR + sl;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = beg; base < end; base += LPR) {
        const int m = min(LPR, end - base);
        int c = g0;
        float w = 0.0f;
        if (sl < m) { c = col[base + sl]; if (MODE == 0) w = dis[c] * di; }
        for (int u = 0; u < m; ++u) {
            const int j = __shfl(c, u, LPR);
            const float wu = __shfl(w, u, LPR);
            const int off = j - g0;
            const float4 v = (tile && (unsigned)off < (unsigned)n_g) ? tile[off * LPR + sl] : xw4g[(size_t)j * D4 + sl];
            if (MODE == 0) fma4(acc, wu, v);
            else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
    }
    if (MODE == 0) {
        fma4(acc, di * di, tile ? tile[(row - g0) * LPR + sl] : xw4g[(size_t)row * D4 + sl]);
        acc.x += bias4.x; acc.y += bias4.y; acc.z += bias4.z; acc.w += bias4.w;
        if (gate) {
            const float4 gt = reinterpret_cast<const float4*>(gate)[o4], hv = reinterpret_cast<const float4*>(h)[o4];
            acc.x = hv.x + gelu_f(acc.x * gt.x); acc.y = hv.y + gelu_f(acc.y * gt.y);
            acc.z = hv.z + gelu_f(acc.z * gt.z); acc.w = hv.w + gelu_f(acc.w * gt.w);
        }
    } else {
        const int cnt = end - beg;
        const float cf = (float)(cnt > 0 ? cnt : 1);
        acc.x /= cf; acc.y /= cf; acc.z /= cf; acc.w /= cf;
    }
    reinterpret_cast<float4*>(out)[o4] = acc;
}

// Every byte of this kernel is touched once: the tile fill (LDS-DMA), the gate / h stream of the epilogue and the output are
// issued NON-TEMPORAL (bit 4 / 2 / 1).  Measured on configs[1] at batch 256, us per launch: 0: 56.6-57.1, 1: 55.9, 2: 57.6,
// 3: 55.3, 4: 55.8, 7: 55.2 (the forward as a whole is unchanged: the next kernel finds h' in HBM either way).
#ifndef AGG_NT
#define AGG_NT 7
#endif
template <int D, int MODE, int SW, bool GATED>
__global__ void __launch_bounds__(AggGraph<SW>::T, AggGraph<SW>::OCC) k_aggregate_graph(int G, const int32_t* __restrict__ node_ptr,
                                                            const float* __restrict__ xw,
                                                            const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ pack,
                                                            const float* __restrict__ dis,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ gate,
                                                            const float* __restrict__ h,
                                                            float* __restrict__ out) {
    using C = AggGraph<SW>;
    constexpr int LPR = C::LPR, RPW = C::RPW, RPP = C::RPP, CAP = C::CAP, K = C::K, FILL = C::FILL, NS = D / SW, D4 = D / 4, T = C::T;
    constexpr int NB = 2;                                  // column batches kept in registers (NB * LPR neighbours)
    constexpr int PF = C::PF;                              // epilogue rows kept in flight ahead of the gather
    constexpr int ZROW = CAP;                              // all-zero tile row (and dis entry) the padding lanes point at
    static_assert(D % SW == 0, "slice width must divide D");
    static_assert(!GATED || MODE == 0, "the gated epilogue belongs to GCNConv");
    extern __shared__ float4 tile[];                       // [CAP + 1][LPR] float4, then dis_l[CAP + 1]
    float* dis_l = reinterpret_cast<float*>(tile + (CAP + 1) * LPR);
    const v4f* tile4 = reinterpret_cast<const v4f*>(tile);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);                    // tells the compiler it is wave-uniform
    const int sub = lane / LPR, sl = lane % LPR;
    const int s = ((int)blockIdx.x >> 3) % NS;                                  // blocks b, b+8, ... share an XCD
    const int g = ((int)blockIdx.x / (8 * NS)) * 8 + ((int)blockIdx.x & 7);
    if (g >= G) return;
    const int g0 = node_ptr[g], n_g = node_ptr[g + 1] - g0;
    if (n_g <= 0) return;
    const float4* xw4g = reinterpret_cast<const float4*>(xw) + s * LPR;         // column slice
    const v4f* gate4 = reinterpret_cast<const v4f*>(gate) + s * LPR + sl;
    const v4f* h4 = reinterpret_cast<const v4f*>(h) + s * LPR + sl;
    const int r0 = wave * RPW + sub;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0 && bias) bias4 = reinterpret_cast<const float4*>(bias)[s * LPR + sl];
    if (n_g > CAP) {                                       // graph larger than the tile: everything from global memory
        for (int r = r0; r < n_g; r += RPP)
            agg_row_generic<D, MODE, SW>(g0 + r, g0, n_g, nullptr, xw4g, row_ptr, col, dis, bias4, gate, h, out, s, sl);
        return;
    }
    // ---- preamble: the packed column words of this block's rows (k_agg_pack) and the tile fill are independent
    // loads, so the block waits for one memory round trip; the gather then touches LDS only.
    int cp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int r = r0 + k * RPP;
        cp[k] = (r < n_g) ? pack[(size_t)(g0 + r) * LPR + sl] : (ZROW | ZROW << 10 | 1 << 20);
    }
    // tile + dis fill: LDS-DMA, no registers; a wave instruction writes 64 consecutive elements
#pragma unroll
    for (int f = 0; f < FILL; ++f) {
        const int i = tid + f * T;
        if (i < n_g * LPR)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xw4g + (size_t)(g0 + i / LPR) * D4 + i % LPR),
                                             (__attribute__((address_space(3))) void*)(tile + f * T + wave * 64), 16, 0, (AGG_NT & 4) ? 2 : 0);
    }
    if (MODE == 0) {
#pragma unroll
        for (int f = 0; f < (CAP + T - 1) / T; ++f) {
            const int i = tid + f * T;
            if (i < n_g)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dis + g0 + i),
                                                 (__attribute__((address_space(3))) void*)(dis_l + f * T + wave * 64), 4, 0, 0);
        }
    }
    if (tid < LPR) tile[ZROW * LPR + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == LPR) dis_l[ZROW] = 0.0f;
    // The gated epilogue streams gate and h from HBM: PF rows ahead are kept in flight, unconditionally (row
    // clamped into the graph) so that the main loop is straight-line code and its waits stay counted.
    v4f gtq[K], hvq[K];
    if (GATED) {
#pragma unroll
        for (int k = 0; k < PF && k < K; ++k) {
            const size_t p4 = (size_t)(g0 + min(r0 + k * RPP, n_g - 1)) * D4;
            if (AGG_NT & 2) { gtq[k] = __builtin_nontemporal_load(&gate4[p4]); hvq[k] = __builtin_nontemporal_load(&h4[p4]); }
            else { gtq[k] = gate4[p4]; hvq[k] = h4[p4]; }
        }
    }
    __syncthreads();
    // ---- gather: LDS only, no global load besides the epilogue prefetch; no predication either, a padding
    // lane adds 0 * 0 from the zero row (acc starts at +0 and x + (+-0) == x, so the sum is unchanged)
    bool irregular = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k * RPP + wave_s * RPW >= n_g) break;          // scalar: no exec-masked region around the body, waits stay counted
        {
            const int r = r0 + k * RPP;
            const bool valid = r < n_g;
            const int rc = valid ? r : n_g - 1;
            if (GATED && k + PF < K) {
                const size_t p4 = (size_t)(g0 + min(r + PF * RPP, n_g - 1)) * D4;
                if (AGG_NT & 2) { gtq[k + PF] = __builtin_nontemporal_load(&gate4[p4]); hvq[k + PF] = __builtin_nontemporal_load(&h4[p4]); }
                else { gtq[k + PF] = gate4[p4]; hvq[k + PF] = h4[p4]; }
            }
            const float di = (MODE == 0) ? dis_l[rc] : 1.0f;
            const int code = cp[k] >> 20;
            const int n = code > 0 ? code - 1 : 0;
            v4f acc = 0.0f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (nb == 0 || nb * LPR < n) {
                    const int myoff = (cp[k] >> (10 * nb)) & 1023;
                    const float myw = (MODE == 0) ? dis_l[myoff] * di : 0.0f;
                    constexpr int UB = C::UB < LPR ? C::UB : LPR;
                    unroll_bcast<1, 0, LPR / UB>([&](auto ub) {
                        v4f v[UB];
                        float wk[UB];
                        unroll_bcast<LPR, 0, UB>([&](auto u) {
                            v[u] = tile4[group_bcast<LPR, ub * UB + u>(myoff) * LPR + sl];
                            if (MODE == 0) wk[u] = group_bcast<LPR, ub * UB + u>(myw);
                        });
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            if (MODE == 0) acc += wk[u] * v[u];
                            else acc += v[u];
                        }
                    });
                }
            }
            if (MODE == 0) {
                acc += (di * di) * tile4[rc * LPR + sl];
                acc += (v4f){bias4.x, bias4.y, bias4.z, bias4.w};
                if (GATED) acc = hvq[k] + gelu4_f(acc * gtq[k]);
            } else {
                acc /= (float)(n > 0 ? n : 1);
            }
            if (valid && code > 0) {
                v4f* o4p = reinterpret_cast<v4f*>(out) + (size_t)(g0 + r) * D4 + s * LPR + sl;
                if (AGG_NT & 1) __builtin_nontemporal_store(acc, o4p); else *o4p = acc;
            }
            irregular = irregular || (valid && code == 0);
        }
    }
    if (__any(irregular)) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int r = r0 + k * RPP;
            if (r < n_g && (cp[k] >> 20) == 0)
                agg_row_generic<D, MODE, SW>(g0 + r, g0, n_g, tile, xw4g, row_ptr, col, dis, bias4, GATED ? gate : nullptr, h, out, s, sl);
        }
    }
}

// ----------------------------------------------------------------- M5: JK fusion
// h_jk = sum_k softmax(jk_logits)_k * states[k]; score = attn . h_jk + b.
template <int D>
__global__ void __launch_bounds__(256) k_jk(int N, int n_states, const float* __restrict__ states,
                                            const float* __restrict__ jk_w /*[n_states] softmaxed*/,
                                            const float* __restrict__ attn_w, const float* __restrict__ attn_b,
                                            float* __restrict__ hjk, float* __restrict__ score) {
    constexpr int LPR = AggCfg<D>::LPR, RPW = AggCfg<D>::RPW, RPB = AggCfg<D>::RPB, D4 = D / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int row = blockIdx.x * RPB + wave * RPW + sub;
    const bool act = row < N && sl * 4 < D;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float dot = 0.0f;
    if (act) {
        const size_t o4 = (size_t)row * D4 + sl;
        const size_t stride4 = (size_t)N * D4;
        for (int k = 0; k < n_states; ++k) {
            const float wk = jk_w[k];
            const float4 v = reinterpret_cast<const float4*>(states)[o4 + stride4 * k];
            acc.x += v.x * wk; acc.y += v.y * wk; acc.z += v.z * wk; acc.w += v.w * wk;
        }
        reinterpret_cast<float4*>(hjk)[o4] = acc;
        const float4 aw = reinterpret_cast<const float4*>(attn_w)[sl];
        dot = ((acc.x * aw.x + acc.y * aw.y) + acc.z * aw.z) + acc.w * aw.w;
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (act && sl == 0) score[row] = dot + attn_b[0];
}

// ------------------------------------------------------ M6: per-graph readout
struct CtxW { const float *wcT /*[D][D/2]*/, *bc, *weT /*[D/2][D]*/, *be; };

template <int D>
__global__ void __launch_bounds__(256) k_graph_ctx(const int32_t* __restrict__ node_ptr,
                                                   const float* __restrict__ score,
                                                   const float* __restrict__ hjk, CtxW w,
                                                   float* __restrict__ gvec) {
    constexpr int Dh = D / 2;
    constexpr int NG = 256 / D;  // column groups (D <= 128 -> >= 2)
    __shared__ float red[256];
    __shared__ float gsum[D];
    __shared__ float cbuf[Dh];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int beg = node_ptr[g], end = node_ptr[g + 1];
    float m = -INFINITY;
    for (int i = beg + tid; i < end; i += 256) m = fmaxf(m, score[i]);
    red[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    m = red[0];
    __syncthreads();
    float s = 0.0f;
    for (int i = beg + tid; i < end; i += 256) s += ggc_expf(score[i] - m);
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float tot = red[0] + 1e-12f;  // model.py:108
    __syncthreads();
    const int d = tid % D, grp = tid / D;
    float acc = 0.0f;
    if (grp < NG)
        for (int i = beg + grp; i < end; i += NG) acc += (ggc_expf(score[i] - m) / tot) * hjk[(size_t)i * D + d];
    red[tid] = (grp < NG) ? acc : 0.0f;
    __syncthreads();
    if (tid < D) {
        float v = 0.0f;
        for (int q = 0; q < NG; ++q) v += red[q * D + tid];
        gsum[tid] = v;
    }
    __syncthreads();
    if (tid < Dh) {
        float a = 0.0f;
        for (int k = 0; k < D; ++k) a += gsum[k] * w.wcT[k * Dh + tid];
        a += w.bc[tid];
        cbuf[tid] = a > 0.0f ? a : 0.0f;
    }
    __syncthreads();
    if (tid < D) {
        float a = 0.0f;
        for (int k = 0; k < Dh; ++k) a += cbuf[k] * w.weT[k * D + tid];
        gvec[(size_t)g * D + tid] = sigmoid_f(a + w.be[tid]);
    }
}

// ------------------------------------------------------------- weight handling

static const float* devp(const ResgcnWeights& m, const std::string& k) {
    auto it = m.dev.find(k);
    return it == m.dev.end() ? nullptr : reinterpret_cast<const float*>(it->second.p);
}
static const float* devp(ggc_ctx* ctx, const std::string& k) { return devp(ctx->model, k); }

static int upload(ggc_ctx* ctx, ResgcnWeights& m, const std::string& key, const std::vector<float>& v);
static int upload(ggc_ctx* ctx, const std::string& key, const std::vector<float>& v) { return upload(ctx, ctx->model, key, v); }
static int upload(ggc_ctx* ctx, ResgcnWeights& m, const std::string& key, const std::vector<float>& v) {
    Buf& b = m.dev[key];
    const size_t bytes = v.size() * sizeof(float);
    if (b.bytes < bytes) {
        if (b.p) GGC_HIP(ctx, hipFree(b.p));
        b.p = nullptr; b.bytes = 0;
        GGC_HIP(ctx, hipMalloc(&b.p, bytes ? bytes : 16));
        b.bytes = bytes;
    }
    if (bytes) GGC_HIP(ctx, hipMemcpy(b.p, v.data(), bytes, hipMemcpyHostToDevice));
    return GGC_OK;
}

static std::vector<float> transpose(const std::vector<float>& w, int out, int in) {
    std::vector<float> t((size_t)out * in);
    for (int o = 0; o < out; ++o)
        for (int k = 0; k < in; ++k) t[(size_t)k * out + o] = w[(size_t)o * in + k];
    return t;
}

// Wp[s/4][t][lane][s%4] = W[32 t + (lane & 31)][(lane >> 5) * D/2 + s]
static std::vector<float> pack_mfma(const std::vector<float>& w, int D) {
    const int T = D / 32, KH = D / 2;
    std::vector<float> p((size_t)D * D);
    for (int s = 0; s < KH; ++s)
        for (int t = 0; t < T; ++t)
            for (int l = 0; l < 64; ++l)
                p[(((size_t)(s / 4) * T + t) * 64 + l) * 4 + (s % 4)] =
                    w[(size_t)(32 * t + (l & 31)) * D + (l >> 5) * KH + s];
    return p;
}

// A state_dict entry: true shape [rows, cols] (cols = 1: a vector) and the shape of its device copy, where every dimension
// that is the hidden width Dt is zero-padded to D (the multiple of 32 the kernels are built for).
struct Need { std::string key; int64_t numel; int r = 0, c = 1, rp = 0, cp = 1; };

static std::vector<Need> needed(const ResgcnWeights& m) {
    const int D = m.D, Dt = m.Dt > 0 ? m.Dt : m.D, Q = m.Q, C = m.C, n = m.n_layers;
    std::vector<Need> v;
    auto add = [&](const std::string& k, int r, int c, int rp, int cp) { v.push_back({k, (int64_t)r * c, r, c, rp, cp}); };
    auto vec = [&](const std::string& k, int len) { add(k, len, 1, len, 1); };
    auto vecD = [&](const std::string& k) { add(k, Dt, 1, D, 1); };
    vec("in_norm.norm.weight", IN_CH); vec("in_norm.norm.bias", IN_CH);
    vec("in_norm.norm.running_mean", IN_CH); vec("in_norm.norm.running_var", IN_CH);
    add("input_proj.0.weight", Dt, IN_CH, D, IN_CH); vecD("input_proj.0.bias");
    vecD("input_proj.1.weight"); vecD("input_proj.1.bias");
    add("prior_booster.0.weight", Q, N_PRIOR, Q, N_PRIOR); vec("prior_booster.0.bias", Q);
    add("prior_booster.2.weight", Dt, Q, D, Q); vecD("prior_booster.2.bias");
    add("edge_ctx.encode.0.weight", C, EDGE_CH, C, EDGE_CH); vec("edge_ctx.encode.0.bias", C);
    add("edge_ctx.encode.2.weight", C, C, C, C); vec("edge_ctx.encode.2.bias", C);
    vec("edge_ctx.to_gate.0.weight", C); vec("edge_ctx.to_gate.0.bias", C);
    add("edge_ctx.to_gate.1.weight", Dt, C, D, C); vecD("edge_ctx.to_gate.1.bias");
    add("sage.lin_l.weight", Dt, Dt, D, D); vecD("sage.lin_l.bias"); add("sage.lin_r.weight", Dt, Dt, D, D);
    vecD("sage_norm.weight"); vecD("sage_norm.bias"); vec("jk_logits", n + 2);
    vecD("ctx.attn.weight"); vec("ctx.attn.bias", 1);
    add("ctx.compress.weight", Dt / 2, Dt, D / 2, D); add("ctx.compress.bias", Dt / 2, 1, D / 2, 1);
    add("ctx.expand.weight", Dt, Dt / 2, D, D / 2); vecD("ctx.expand.bias");
    vecD("fuse.0.weight"); vecD("fuse.0.bias"); add("fuse.1.weight", Dt, Dt, D, D); vecD("fuse.1.bias");
    add("head.weight", N_CLS, Dt, N_CLS, D); vec("head.bias", N_CLS);
    for (int i = 0; i < n; ++i) {
        const std::string s = std::to_string(i);
        vecD("gcn_layers." + s + ".bias");
        add("gcn_layers." + s + ".lin.weight", Dt, Dt, D, D);
        vecD("norms." + s + ".weight");
        vecD("norms." + s + ".bias");
    }
    return v;
}

// zero-padded copy [rp, cp] of a row-major [r, c] array
static std::vector<float> pad2(const std::vector<float>& w, const Need& nd) {
    if (nd.r == nd.rp && nd.c == nd.cp) return w;
    std::vector<float> p((size_t)nd.rp * nd.cp, 0.0f);
    for (int i = 0; i < nd.r; ++i)
        for (int j = 0; j < nd.c; ++j) p[(size_t)i * nd.cp + j] = w[(size_t)i * nd.c + j];
    return p;
}

static int check_ready(ggc_ctx* ctx) {
    ResgcnWeights& m = ctx->model;
    GGC_REQUIRE(ctx, m.D > 0, GGC_E_STATE, "ggc_resgcn_configure has not been called");
    for (const Need& nd : needed(m)) {
        auto it = m.host.find(nd.key);
        GGC_REQUIRE(ctx, it != m.host.end(), GGC_E_STATE, "missing weight '%s'", nd.key.c_str());
        GGC_REQUIRE(ctx, (int64_t)it->second.size() == nd.numel, GGC_E_SHAPE,
                    "weight '%s' has %zu elements, expected %lld", nd.key.c_str(), it->second.size(),
                    (long long)nd.numel);
    }
    return GGC_OK;
}

static int prepare_weights(ggc_ctx* ctx) {
    ResgcnWeights& m = ctx->model;
    if (m.dev_ok) return GGC_OK;
    int rc = check_ready(ctx);
    if (rc) return rc;
    // The device copies are about to be overwritten in place by blocking copies on the null stream, which does not wait for
    // the (non-blocking) stream a previous forward may still be running on: drain the device first.  Weight changes are rare.
    GGC_HIP(ctx, hipDeviceSynchronize());
    const int D = m.D, Q = m.Q, C = m.C, n = m.n_layers;
    // device copies are the zero-padded arrays (a width that is a multiple of 32 pads nothing); m.host keeps what was loaded
    std::map<std::string, std::vector<float>> pw;
    for (const Need& nd : needed(m)) pw[nd.key] = pad2(m.host[nd.key], nd);
    for (auto& kv : pw) { rc = upload(ctx, kv.first, kv.second); if (rc) return rc; }
    if ((rc = upload(ctx, "#input_proj.0.weightT", transpose(pw["input_proj.0.weight"], D, IN_CH)))) return rc;
    if ((rc = upload(ctx, "#prior_booster.2.weightT", transpose(pw["prior_booster.2.weight"], D, Q)))) return rc;
    if ((rc = upload(ctx, "#edge_ctx.encode.2.weightT", transpose(pw["edge_ctx.encode.2.weight"], C, C)))) return rc;
    if ((rc = upload(ctx, "#edge_ctx.to_gate.1.weightT", transpose(pw["edge_ctx.to_gate.1.weight"], D, C)))) return rc;
    if ((rc = upload(ctx, "#ctx.compress.weightT", transpose(pw["ctx.compress.weight"], D / 2, D)))) return rc;
    if ((rc = upload(ctx, "#ctx.expand.weightT", transpose(pw["ctx.expand.weight"], D, D / 2)))) return rc;
    for (int i = 0; i < n; ++i) {
        const std::string k = "gcn_layers." + std::to_string(i) + ".lin.weight";
        if ((rc = upload(ctx, "#" + k + ".p", pack_mfma(pw[k], D)))) return rc;
    }
    if ((rc = upload(ctx, "#sage.lin_l.weight.p", pack_mfma(pw["sage.lin_l.weight"], D)))) return rc;
    if ((rc = upload(ctx, "#sage.lin_r.weight.p", pack_mfma(pw["sage.lin_r.weight"], D)))) return rc;
    if ((rc = upload(ctx, "#fuse.1.weight.p", pack_mfma(pw["fuse.1.weight"], D)))) return rc;
    {   // softmax(jk_logits) on the host (model.py:532), same op order as the oracle
        const std::vector<float>& jl = m.host["jk_logits"];
        std::vector<float> w(jl.size());
        float mx = jl[0];
        for (float v : jl) mx = v > mx ? v : mx;
        float s = 0.0f;
        for (size_t k = 0; k < jl.size(); ++k) { w[k] = ggc_expf(jl[k] - mx); s += w[k]; }
        for (size_t k = 0; k < jl.size(); ++k) w[k] = w[k] / s;
        if ((rc = upload(ctx, "#jk_w", w))) return rc;
    }
    m.dev_ok = true;
    return GGC_OK;
}

// ------------------------------------------------------------ launch helpers

template <int D, int MODE>
static int launch_gemm(ggc_ctx* ctx, hipStream_t st, int N, const GemmArgs& a) {
    static DeviceOnce attr_set;                    // per device; contexts may live on other host threads
    const size_t lds = (size_t)D * D * sizeof(float);
    if (attr_set.need(ctx->device) && lds > 48 * 1024) {
        GGC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<D, MODE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done(ctx->device);
    }
    ProfScope prof(ctx, st, MODE == 0 ? "gcn_gemm" : MODE == 1 ? "sage_gemm" : (MODE == 2 || MODE == 4) ? "head_gemm" : "plain_gemm");
    hipLaunchKernelGGL((k_gemm<D, MODE>), dim3(cdiv(N, 128)), dim3(256), lds, st, N, a);
    GGC_LAUNCH_CHECK(ctx);
#ifdef GEMM_TRACE
    if (D == 128 && MODE == 0 && std::getenv("GGC_GEMM_TRACE")) {
        static std::vector<unsigned long long> host(8192 * 8);
        int occ = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k_gemm<D, MODE>), 256, lds);
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(g_gemm_trace), host.size() * 8);
        if (FILE* f = std::fopen(std::getenv("GGC_GEMM_TRACE"), "ab")) {
            const long long n = std::min(cdiv(N, 128) * 4, 8192);
            std::fwrite(&n, 8, 1, f); std::fwrite(host.data(), 8, (size_t)n * 8, f); std::fclose(f);
        }
        std::fprintf(stderr, "k_gemm<128,0> trace: blocks per CU %d\n", occ);
    }
#endif
    return GGC_OK;
}

template <int D, int MODE, int SW, bool GATED>
static int launch_aggregate_graph_t(ggc_ctx* ctx, hipStream_t st, int G, const int32_t* node_ptr, const int32_t* pack,
                                    const float* xw, const int32_t* row_ptr, const int32_t* col, const float* dis,
                                    const float* bias, const float* gate, const float* h, float* out) {
    static DeviceOnce attr_set;
    if (attr_set.need(ctx->device)) {
        GGC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aggregate_graph<D, MODE, SW, GATED>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, AggGraph<SW>::LDS));
        attr_set.done(ctx->device);
    }
    hipLaunchKernelGGL((k_aggregate_graph<D, MODE, SW, GATED>), dim3(cdiv(G, 8) * 8 * (D / SW)), dim3(AggGraph<SW>::T), AggGraph<SW>::LDS, st,
                       G, node_ptr, xw, row_ptr, col, pack, dis, bias, gate, h, out);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

// Column-slice width of the graph-resident gather for G graphs with N nodes in total: 32 when the [nodes-per-graph, 32]
// tile fits half a CU's LDS with a little head-room over the mean graph size (a larger graph falls back per block);
// 0 = use the direct gather.
static int agg_graph_slice(int N, int G) {
    if (G <= 0 || knobs().agg_direct) return 0;
    const double want = 1.02 * (double)N / G;
    return want <= AggGraph<32>::CAP ? 32 : 0;
}

// The forward pass hands over the batch structure (G graphs, node_ptr) and the packed column words built by
// build_agg_pack for the slice width agg_graph_slice chose; without them (ggc_gcn_aggregate) the direct gather runs.
struct AggGraphs { int G = 0, sw = 0; const int32_t* node_ptr = nullptr; const int32_t* pack = nullptr; };

static int build_agg_pack(ggc_ctx* ctx, hipStream_t st, int N, int G, const int32_t* node_ptr, const int32_t* batch,
                          const int32_t* row_ptr, const int32_t* col, AggGraphs& ag) {
    ag = AggGraphs{};
    const int sw = agg_graph_slice(N, G);
    if (!sw || N <= 0) return GGC_OK;
    const int lpr = sw / 4;
    int32_t* pack = scratch_t<int32_t>(ctx, S_AGG_PACK, (size_t)N * lpr);
    if (!pack) return GGC_E_OOM;
    hipLaunchKernelGGL(k_agg_pack<32>, dim3(cdiv(N * lpr, 256)), dim3(256), 0, st, N, node_ptr, batch, row_ptr, col, pack);
    GGC_LAUNCH_CHECK(ctx);
    ag.G = G; ag.sw = sw; ag.node_ptr = node_ptr; ag.pack = pack;
    return GGC_OK;
}

template <int D, int MODE, int SW>
static int launch_aggregate_graph(ggc_ctx* ctx, hipStream_t st, const AggGraphs& ag, const float* xw,
                                  const int32_t* row_ptr, const int32_t* col, const float* dis, const float* bias,
                                  const float* gate, const float* h, float* out) {
    if constexpr (MODE == 0) {
        if (gate)
            return launch_aggregate_graph_t<D, MODE, SW, true>(ctx, st, ag.G, ag.node_ptr, ag.pack, xw, row_ptr, col, dis, bias, gate, h, out);
    }
    return launch_aggregate_graph_t<D, MODE, SW, false>(ctx, st, ag.G, ag.node_ptr, ag.pack, xw, row_ptr, col, dis, bias, nullptr, nullptr, out);
}

template <int D, int MODE>
static int launch_aggregate(ggc_ctx* ctx, hipStream_t st, int N, const float* xw, const int32_t* row_ptr,
                            const int32_t* col, const float* dis, const float* bias, const float* gate,
                            const float* h, float* out, const AggGraphs& ag = AggGraphs{}) {
    ProfScope prof(ctx, st, MODE == 0 ? "gcn_aggregate" : "sage_aggregate");
    if (ag.sw == 32) return launch_aggregate_graph<D, MODE, 32>(ctx, st, ag, xw, row_ptr, col, dis, bias, gate, h, out);
    constexpr int threads = 256;
    const int rows_per_block = AggCfg<D>::RPW * (threads / 64);
    hipLaunchKernelGGL((k_aggregate<D, MODE>), dim3(cdiv(N, rows_per_block)), dim3(threads), 0, st,
                       N, xw, row_ptr, col, dis, bias, gate, h, out);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

template <int D>
static int forward_t(ggc_ctx* ctx, hipStream_t st, int G, int N, int E, const float* x,
                     const int32_t* edge_src, const int32_t* edge_dst, const float* edge_attr,
                     const int32_t* node_ptr, float* logits, float* probs) {
    ResgcnWeights& m = ctx->model;
    const int n = m.n_layers, n_states = n + 2;
    const size_t ND = (size_t)N * D;
    int32_t* row_ptr = scratch_t<int32_t>(ctx, S_CSR_ROWPTR, (size_t)N + 1);
    int32_t* col = scratch_t<int32_t>(ctx, S_CSR_COL, (size_t)E);
    int32_t* eid = scratch_t<int32_t>(ctx, S_CSR_EID, (size_t)E);
    int32_t* cursor = scratch_t<int32_t>(ctx, S_CSR_CURSOR, (size_t)N + 1);
    float* dis = scratch_t<float>(ctx, S_DIS, (size_t)N);
    int32_t* batch = scratch_t<int32_t>(ctx, S_BATCH, (size_t)N);
    float* states = scratch_t<float>(ctx, S_STATES, ND * n_states);
    float* gate = scratch_t<float>(ctx, S_GATE, ND);
    float* xw = scratch_t<float>(ctx, S_XW, ND);
    float* agg = scratch_t<float>(ctx, S_AGG, ND);
    float* hjk = scratch_t<float>(ctx, S_HJK, ND);
    float* score = scratch_t<float>(ctx, S_SCORE, (size_t)N);
    float* gvec = scratch_t<float>(ctx, S_GVEC, (size_t)G * D);
    if (!row_ptr || !col || !eid || !cursor || !dis || !batch || !states || !gate || !xw || !agg || !hjk ||
        !score || !gvec)
        return GGC_E_OOM;

    int rc = build_csr(ctx, st, N, E, edge_src, edge_dst, row_ptr, col, eid, cursor, dis);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fill_batch, dim3(min(cdiv(N, 256), 4096)), dim3(256), 0, st, G, N, node_ptr, batch);
    GGC_LAUNCH_CHECK(ctx);
    AggGraphs ag;
    if ((rc = build_agg_pack(ctx, st, N, G, node_ptr, batch, row_ptr, col, ag))) return rc;

    const int wave_blocks = min(cdiv(N, 4), 8 * ctx->n_cu);
    {
        InputW w{devp(ctx, "in_norm.norm.weight"), devp(ctx, "in_norm.norm.bias"),
                 devp(ctx, "in_norm.norm.running_mean"), devp(ctx, "in_norm.norm.running_var"),
                 devp(ctx, "#input_proj.0.weightT"), devp(ctx, "input_proj.0.bias"),
                 devp(ctx, "input_proj.1.weight"), devp(ctx, "input_proj.1.bias"),
                 devp(ctx, "prior_booster.0.weight"), devp(ctx, "prior_booster.0.bias"),
                 devp(ctx, "#prior_booster.2.weightT"), devp(ctx, "prior_booster.2.bias")};
        hipLaunchKernelGGL((k_input<D>), dim3(wave_blocks), dim3(256), 0, st, N, x, w, m.Q, m.Dt, states);
        GGC_LAUNCH_CHECK(ctx);
    }
    {
        EdgeW w{devp(ctx, "edge_ctx.encode.0.weight"), devp(ctx, "edge_ctx.encode.0.bias"),
                devp(ctx, "#edge_ctx.encode.2.weightT"), devp(ctx, "edge_ctx.encode.2.bias"),
                devp(ctx, "edge_ctx.to_gate.0.weight"), devp(ctx, "edge_ctx.to_gate.0.bias"),
                devp(ctx, "#edge_ctx.to_gate.1.weightT"), devp(ctx, "edge_ctx.to_gate.1.bias")};
        const size_t eg_lds = sizeof(float) * ((size_t)m.C * m.C + (size_t)m.C * D);            // <= 48 KB (C <= 64, D <= 128)
        hipLaunchKernelGGL((k_edge_gate<D>), dim3(min(cdiv(N, 4 * EC_NODES), 3 * ctx->n_cu)), dim3(256), eg_lds, st, N, row_ptr, eid,
                           edge_attr, w, m.C, gate);
        GGC_LAUNCH_CHECK(ctx);
    }
    for (int l = 0; l < n; ++l) {
        const std::string s = std::to_string(l);
        const float* h_in = states + ND * l;
        float* h_out = states + ND * (l + 1);
        GemmArgs a{};
        a.A1 = h_in; a.Wp1 = devp(ctx, "#gcn_layers." + s + ".lin.weight.p");
        a.ln_w = devp(ctx, "norms." + s + ".weight"); a.ln_b = devp(ctx, "norms." + s + ".bias");
        a.out = xw; a.Dt = m.Dt;
        if ((rc = launch_gemm<D, 0>(ctx, st, N, a))) return rc;
        if ((rc = launch_aggregate<D, 0>(ctx, st, N, xw, row_ptr, col, dis, devp(ctx, "gcn_layers." + s + ".bias"),
                                         gate, h_in, h_out, ag)))
            return rc;
    }
    {
        const float* hl = states + ND * n;
        if ((rc = launch_aggregate<D, 1>(ctx, st, N, hl, row_ptr, col, nullptr, nullptr, nullptr, nullptr, agg, ag)))
            return rc;
        GemmArgs a{};
        a.A1 = agg; a.A2 = hl;
        a.Wp1 = devp(ctx, "#sage.lin_l.weight.p"); a.Wp2 = devp(ctx, "#sage.lin_r.weight.p");
        a.bias = devp(ctx, "sage.lin_l.bias");
        a.ep_w = devp(ctx, "sage_norm.weight"); a.ep_b = devp(ctx, "sage_norm.bias");
        a.out = states + ND * (n + 1); a.Dt = m.Dt;
        if ((rc = launch_gemm<D, 1>(ctx, st, N, a))) return rc;
    }
    hipLaunchKernelGGL((k_jk<D>), dim3(cdiv(N, AggCfg<D>::RPB)), dim3(256), 0, st, N, n_states, states,
                       devp(ctx, "#jk_w"), devp(ctx, "ctx.attn.weight"), devp(ctx, "ctx.attn.bias"), hjk, score);
    GGC_LAUNCH_CHECK(ctx);
    {
        CtxW w{devp(ctx, "#ctx.compress.weightT"), devp(ctx, "ctx.compress.bias"),
               devp(ctx, "#ctx.expand.weightT"), devp(ctx, "ctx.expand.bias")};
        hipLaunchKernelGGL((k_graph_ctx<D>), dim3(G), dim3(256), 0, st, node_ptr, score, hjk, w, gvec);
        GGC_LAUNCH_CHECK(ctx);
    }
    {
        GemmArgs a{};
        a.A1 = hjk; a.Wp1 = devp(ctx, "#fuse.1.weight.p");
        a.ln_w = devp(ctx, "fuse.0.weight"); a.ln_b = devp(ctx, "fuse.0.bias");
        a.bias = devp(ctx, "fuse.1.bias");
        a.batch = batch; a.gvec = gvec;
        a.ep_w = devp(ctx, "head.weight"); a.ep_b = devp(ctx, "head.bias");
        a.out = logits; a.out2 = probs; a.Dt = m.Dt;
        if ((rc = launch_gemm<D, 2>(ctx, st, N, a))) return rc;
    }
    return GGC_OK;
}

} // namespace ggc

using namespace ggc;

extern "C" {

int ggc_resgcn_configure(ggc_ctx* ctx, int hidden, int n_layers) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, hidden >= 8 && hidden <= 128, GGC_E_UNSUPPORTED,
                "hidden_channels=%d unsupported: the kernels are built for widths from 8 to 128", hidden);
    GGC_REQUIRE(ctx, n_layers >= 1 && n_layers <= 30, GGC_E_INVALID_ARG, "n_layers=%d out of range [1,30]", n_layers);
    ResgcnWeights& m = ctx->model;
    if (m.Dt != hidden || m.n_layers != n_layers) { m.host.clear(); }
    // widths that are not a multiple of 32 run zero-padded to the next one (k_input: only the LayerNorm statistics see Dt)
    m.Dt = hidden; m.D = (hidden + 31) / 32 * 32; m.n_layers = n_layers;
    m.Q = hidden / 4 > 8 ? hidden / 4 : 8;   // model.py:472
    m.C = hidden / 2 > 8 ? hidden / 2 : 8;   // model.py:123
    m.dev_ok = false;
    return GGC_OK;
}

int ggc_resgcn_load_weight(ggc_ctx* ctx, const char* name, const float* data, int64_t numel) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, name && (data || numel == 0) && numel >= 0, GGC_E_INVALID_ARG, "bad weight arguments");
    GGC_REQUIRE(ctx, ctx->model.D > 0, GGC_E_STATE, "ggc_resgcn_configure has not been called");
    const std::string key(name);
    const std::string tail = "num_batches_tracked";
    if (key.size() >= tail.size() && key.compare(key.size() - tail.size(), tail.size(), tail) == 0) return GGC_OK;
    bool known = false;
    for (const Need& nd : needed(ctx->model))
        if (nd.key == key) {
            GGC_REQUIRE(ctx, nd.numel == numel, GGC_E_SHAPE, "weight '%s' has %lld elements, expected %lld", name,
                        (long long)numel, (long long)nd.numel);
            known = true;
            break;
        }
    GGC_REQUIRE(ctx, known, GGC_E_INVALID_ARG, "unexpected state_dict key '%s' for ResGCNNet(D=%d, n=%d)", name,
                ctx->model.Dt, ctx->model.n_layers);
    ctx->model.host[key].assign(data, data + numel);
    ctx->model.dev_ok = false;
    return GGC_OK;
}

int ggc_resgcn_ready(ggc_ctx* ctx) {
    if (!ctx) return GGC_E_INVALID_ARG;
    return check_ready(ctx);
}

int ggc_resgcn_forward(ggc_ctx* ctx, ggc_stream stream, int G, int N, int E, const float* x,
                       const int32_t* edge_src, const int32_t* edge_dst, const float* edge_attr,
                       const int32_t* node_ptr, float* logits, float* probs) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, G >= 1 && N >= 1 && E >= 0, GGC_E_SHAPE, "bad sizes G=%d N=%d E=%d", G, N, E);
    GGC_REQUIRE(ctx, x && node_ptr && (E == 0 || (edge_src && edge_dst && edge_attr)), GGC_E_INVALID_ARG,
                "null input pointer");
    GGC_REQUIRE(ctx, logits || probs, GGC_E_INVALID_ARG, "both outputs are NULL");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    int rc = prepare_weights(ctx);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (ctx->model.D) {
        case 32:  return forward_t<32>(ctx, st, G, N, E, x, edge_src, edge_dst, edge_attr, node_ptr, logits, probs);
        case 64:  return forward_t<64>(ctx, st, G, N, E, x, edge_src, edge_dst, edge_attr, node_ptr, logits, probs);
        case 96:  return forward_t<96>(ctx, st, G, N, E, x, edge_src, edge_dst, edge_attr, node_ptr, logits, probs);
        case 128: return forward_t<128>(ctx, st, G, N, E, x, edge_src, edge_dst, edge_attr, node_ptr, logits, probs);
    }
    return set_err(ctx, GGC_E_UNSUPPORTED, "hidden=%d", ctx->model.D);
}

int ggc_build_csr(ggc_ctx* ctx, ggc_stream stream, int N, int E, const int32_t* edge_src,
                  const int32_t* edge_dst, int32_t* row_ptr, int32_t* col, float* dis) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, N >= 1 && E >= 0 && row_ptr && (E == 0 || (edge_src && edge_dst && col)), GGC_E_INVALID_ARG,
                "bad arguments");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    int32_t* eid = scratch_t<int32_t>(ctx, S_CSR_EID, (size_t)E);
    int32_t* cursor = scratch_t<int32_t>(ctx, S_CSR_CURSOR, (size_t)N + 1);
    if (!eid || !cursor) return GGC_E_OOM;
    return build_csr(ctx, reinterpret_cast<hipStream_t>(stream), N, E, edge_src, edge_dst, row_ptr, col, eid,
                     cursor, dis);
}

int ggc_gcn_aggregate(ggc_ctx* ctx, ggc_stream stream, int N, int D, const float* xw,
                      const int32_t* row_ptr, const int32_t* col, const float* dis, const float* bias,
                      const float* gate, const float* h, float* h_out) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, N >= 1 && xw && row_ptr && col && dis && h_out, GGC_E_INVALID_ARG, "bad arguments");
    GGC_REQUIRE(ctx, (gate == nullptr) == (h == nullptr), GGC_E_INVALID_ARG, "gate and h must be given together");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (D) {
        case 32:  return launch_aggregate<32, 0>(ctx, st, N, xw, row_ptr, col, dis, bias, gate, h, h_out);
        case 64:  return launch_aggregate<64, 0>(ctx, st, N, xw, row_ptr, col, dis, bias, gate, h, h_out);
        case 96:  return launch_aggregate<96, 0>(ctx, st, N, xw, row_ptr, col, dis, bias, gate, h, h_out);
        case 128: return launch_aggregate<128, 0>(ctx, st, N, xw, row_ptr, col, dis, bias, gate, h, h_out);
    }
    return set_err(ctx, GGC_E_UNSUPPORTED, "D=%d unsupported (32, 64, 96, 128)", D);
}

} // extern "C"

// ===================================================================================================
// GCNTrimapNet (reference model.py:239-316), eval mode.
//   in_norm -> Linear(19, D) + BatchNorm + ReLU -> n x ResGCNBlock -> head on the concatenated block outputs
//   ResGCNBlock (:216-232): h' = (relu(bn(GCNConv(h))) + h) * scatter_mean_dst(sigmoid(W2 relu(W1 e + b1) + b2))
// Reuses the destination CSR, the f32-MFMA product (k_gemm mode 3, no prologue norm) and the GCNConv gather of the
// ResGCNNet path.  The per-edge gate MLP, its scatter-mean and the block's BatchNorm / ReLU / residual epilogue are one
// kernel (k_gn_edge_gate, at the end of this file).  BatchNorm1d(eval) = (x - mean) / sqrt(var + 1e-5) * w + b.
// ===================================================================================================
namespace ggc {

// destination (row) of every CSR position: row_ptr[r] <= j < row_ptr[r + 1]
__global__ void __launch_bounds__(256) k_csr_dst(int N, int E, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ csr_dst) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E) return;
    int lo = 0, hi = N;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row_ptr[mid] <= j) lo = mid; else hi = mid; }
    csr_dst[j] = lo;
}

struct BnW { const float *w, *b, *rm, *rv; };
__device__ __forceinline__ float bn_apply(float x, const BnW& p, int k) {
    return (x - p.rm[k]) / sqrtf(p.rv[k] + 1e-5f) * p.w[k] + p.b[k];
}

// in_norm + input_proj: one wave per node
template <int D>
__global__ void __launch_bounds__(256) k_gn_input(int N, const float* __restrict__ x, BnW bn_in, const float* __restrict__ w_inT,
                                                  const float* __restrict__ b_in, BnW bn1, float* __restrict__ h) {
    constexpr int NC = (D + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int node = wave; node < N; node += n_waves) {
        float xn[IN_CH];
#pragma unroll
        for (int k = 0; k < IN_CH; ++k) xn[k] = bn_apply(x[(size_t)node * IN_CH + k], bn_in, k);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < IN_CH; ++k) acc += xn[k] * w_inT[k * D + c];
                const float v = bn_apply(acc + b_in[c], bn1, c);
                h[(size_t)node * D + c] = v > 0.0f ? v : 0.0f;
            }
        }
    }
}

// head tail: z = relu(bn(z0 + b0)) -> Linear(D, D/2) + ReLU -> Linear(D/2, 3) (+ softmax); one wave per node
template <int D>
__global__ void __launch_bounds__(256) k_gn_head(int N, const float* __restrict__ z0, const float* __restrict__ b0, BnW bn,
                                                 const float* __restrict__ w4T /*[D][D/2]*/, const float* __restrict__ b4,
                                                 const float* __restrict__ w6 /*[3][D/2]*/, const float* __restrict__ b6,
                                                 float* __restrict__ logits, float* __restrict__ probs) {
    constexpr int DH = D / 2, NC = (D + 63) / 64;
    __shared__ float s_z[4][D];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int node = wave; node < N; node += n_waves) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                const float v = bn_apply(z0[(size_t)node * D + c] + b0[c], bn, c);
                s_z[wv][c] = v > 0.0f ? v : 0.0f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float a = 0.0f;                                    // hidden unit `lane` of the D/2 layer (D/2 <= 64)
        if (lane < DH) {
            for (int k = 0; k < D; ++k) a += s_z[wv][k] * w4T[k * DH + lane];
            a += b4[lane];
            a = a > 0.0f ? a : 0.0f;
        }
        float p[N_CLS];
#pragma unroll
        for (int c = 0; c < N_CLS; ++c) {
            // index-order sum like the oracle: lane k contributes a_k * w6[c][k], folded sequentially by lane 0
            p[c] = (lane < DH) ? a * w6[c * DH + lane] : 0.0f;
        }
        float lg[N_CLS] = {0.0f, 0.0f, 0.0f};
        for (int k = 0; k < DH; ++k) {
#pragma unroll
            for (int c = 0; c < N_CLS; ++c) lg[c] += __shfl(p[c], k, 64);
        }
        if (lane == 0) {
            const float l0 = lg[0] + b6[0], l1 = lg[1] + b6[1], l2 = lg[2] + b6[2];
            if (logits) { logits[(size_t)node * 3 + 0] = l0; logits[(size_t)node * 3 + 1] = l1; logits[(size_t)node * 3 + 2] = l2; }
            if (probs) {
                const float mx = fmaxf(l0, fmaxf(l1, l2));
                const float e0 = ggc_expf(l0 - mx), e1 = ggc_expf(l1 - mx), e2 = ggc_expf(l2 - mx);
                const float s = (e0 + e1) + e2;
                probs[(size_t)node * 3 + 0] = e0 / s; probs[(size_t)node * 3 + 1] = e1 / s; probs[(size_t)node * 3 + 2] = e2 / s;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static const char* const BN_KEYS[4] = {"weight", "bias", "running_mean", "running_var"};

static std::vector<Need> needed_gcnnet(const ResgcnWeights& m) {
    const int D = m.D, n = m.n_layers;
    std::vector<Need> v;
    for (const char* k : BN_KEYS) v.push_back({std::string("in_norm.norm.") + k, IN_CH});
    v.push_back({"input_proj.0.weight", (int64_t)D * IN_CH}); v.push_back({"input_proj.0.bias", D});
    for (const char* k : BN_KEYS) v.push_back({std::string("input_proj.1.") + k, D});
    for (int i = 0; i < n; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        v.push_back({p + "conv.bias", D}); v.push_back({p + "conv.lin.weight", (int64_t)D * D});
        for (const char* k : BN_KEYS) v.push_back({p + "bn." + k, D});
        v.push_back({p + "edge_inject.proj.0.weight", (int64_t)D * EDGE_CH}); v.push_back({p + "edge_inject.proj.0.bias", D});
        v.push_back({p + "edge_inject.proj.2.weight", (int64_t)D * D}); v.push_back({p + "edge_inject.proj.2.bias", D});
    }
    v.push_back({"head.0.weight", (int64_t)D * D * (n + 1)}); v.push_back({"head.0.bias", D});
    for (const char* k : BN_KEYS) v.push_back({std::string("head.1.") + k, D});
    v.push_back({"head.4.weight", (int64_t)(D / 2) * D}); v.push_back({"head.4.bias", D / 2});
    v.push_back({"head.6.weight", (int64_t)N_CLS * (D / 2)}); v.push_back({"head.6.bias", N_CLS});
    return v;
}

static int check_ready_gcnnet(ggc_ctx* ctx) {
    ResgcnWeights& m = ctx->model2;
    GGC_REQUIRE(ctx, m.D > 0, GGC_E_STATE, "ggc_gcnnet_configure has not been called");
    for (const Need& nd : needed_gcnnet(m)) {
        auto it = m.host.find(nd.key);
        GGC_REQUIRE(ctx, it != m.host.end(), GGC_E_STATE, "missing weight '%s'", nd.key.c_str());
        GGC_REQUIRE(ctx, (int64_t)it->second.size() == nd.numel, GGC_E_SHAPE, "weight '%s' has %zu elements, expected %lld",
                    nd.key.c_str(), it->second.size(), (long long)nd.numel);
    }
    return GGC_OK;
}

static int prepare_weights_gcnnet(ggc_ctx* ctx) {
    ResgcnWeights& m = ctx->model2;
    if (m.dev_ok) return GGC_OK;
    int rc = check_ready_gcnnet(ctx);
    if (rc) return rc;
    GGC_HIP(ctx, hipDeviceSynchronize());                  // (see prepare_weights)
    const int D = m.D, n = m.n_layers;
    for (auto& kv : m.host) { if ((rc = upload(ctx, m, kv.first, kv.second))) return rc; }
    if ((rc = upload(ctx, m, "#input_proj.0.weightT", transpose(m.host["input_proj.0.weight"], D, IN_CH)))) return rc;
    if ((rc = upload(ctx, m, "#head.4.weightT", transpose(m.host["head.4.weight"], D / 2, D)))) return rc;
    for (int i = 0; i < n; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        if ((rc = upload(ctx, m, "#" + p + "conv.lin.weight.p", pack_mfma(m.host[p + "conv.lin.weight"], D)))) return rc;
        if ((rc = upload(ctx, m, "#" + p + "edge_inject.proj.0.weightT", transpose(m.host[p + "edge_inject.proj.0.weight"], D, EDGE_CH)))) return rc;
        if ((rc = upload(ctx, m, "#" + p + "edge_inject.proj.2.weight.p", pack_mfma(m.host[p + "edge_inject.proj.2.weight"], D)))) return rc;
    }
    const std::vector<float>& hw = m.host["head.0.weight"];       // [D][D (n+1)]: one D x D block per concatenated state
    for (int s = 0; s <= n; ++s) {
        std::vector<float> blk((size_t)D * D);
        for (int o = 0; o < D; ++o)
            for (int k = 0; k < D; ++k) blk[(size_t)o * D + k] = hw[(size_t)o * D * (n + 1) + (size_t)s * D + k];
        if ((rc = upload(ctx, m, "#head.0.weight.p" + std::to_string(s), pack_mfma(blk, D)))) return rc;
    }
    m.dev_ok = true;
    return GGC_OK;
}

static BnW bn_of(const ResgcnWeights& m, const std::string& prefix) {
    return BnW{devp(m, prefix + "weight"), devp(m, prefix + "bias"), devp(m, prefix + "running_mean"), devp(m, prefix + "running_var")};
}

template <int D, bool MUL_ONLY>
static int launch_edge_gate(ggc_ctx* ctx, hipStream_t st, int N, const int32_t* row_ptr, const int32_t* eid, const int32_t* csr_dst,
                            const float* edge_attr, const float* w1T, const float* b1, const float* w2p, const float* b2,
                            const float* conv, const BnW& bn, const float* h, float* out);

template <int D>
static int forward_gcnnet_t(ggc_ctx* ctx, hipStream_t st, int N, int E, const float* x, const int32_t* edge_src,
                            const int32_t* edge_dst, const float* edge_attr, float* logits, float* probs) {
    ResgcnWeights& m = ctx->model2;
    const int n = m.n_layers, n_states = n + 1;
    const size_t ND = (size_t)N * D;
    int32_t* row_ptr = scratch_t<int32_t>(ctx, S_CSR_ROWPTR, (size_t)N + 1);
    int32_t* col = scratch_t<int32_t>(ctx, S_CSR_COL, (size_t)std::max(E, 1));
    int32_t* eid = scratch_t<int32_t>(ctx, S_CSR_EID, (size_t)std::max(E, 1));
    int32_t* cursor = scratch_t<int32_t>(ctx, S_CSR_CURSOR, (size_t)N + 1);
    float* dis = scratch_t<float>(ctx, S_DIS, (size_t)N);
    float* states = scratch_t<float>(ctx, S_STATES, ND * n_states);
    float* xw = scratch_t<float>(ctx, S_XW, ND);
    float* conv = scratch_t<float>(ctx, S_AGG, ND);
    float* z0 = scratch_t<float>(ctx, S_HJK, ND);
    if (!row_ptr || !col || !eid || !cursor || !dis || !states || !xw || !conv || !z0) return GGC_E_OOM;
    int rc = build_csr(ctx, st, N, E, edge_src, edge_dst, row_ptr, col, eid, cursor, dis);
    if (rc) return rc;
    int32_t* csr_dst = scratch_t<int32_t>(ctx, S_AGG_PACK, (size_t)std::max(E, 1));        // destination of every CSR position
    if (!csr_dst) return GGC_E_OOM;
    if (E > 0) {
        hipLaunchKernelGGL(k_csr_dst, dim3(cdiv(E, 256)), dim3(256), 0, st, N, E, row_ptr, csr_dst);
        GGC_LAUNCH_CHECK(ctx);
    }
    const int wave_blocks = min(cdiv(N, 4), 8 * ctx->n_cu);
    hipLaunchKernelGGL((k_gn_input<D>), dim3(wave_blocks), dim3(256), 0, st, N, x, bn_of(m, "in_norm.norm."),
                       devp(m, "#input_proj.0.weightT"), devp(m, "input_proj.0.bias"), bn_of(m, "input_proj.1."), states);
    GGC_LAUNCH_CHECK(ctx);
    for (int l = 0; l < n; ++l) {
        const std::string p = "blocks." + std::to_string(l) + ".";
        const float* h = states + ND * l;
        float* out = states + ND * (l + 1);
        GemmArgs a{};
        a.A1 = h; a.Wp1 = devp(m, "#" + p + "conv.lin.weight.p"); a.out = xw;
        if ((rc = launch_gemm<D, 3>(ctx, st, N, a))) return rc;
        if ((rc = launch_aggregate<D, 0>(ctx, st, N, xw, row_ptr, col, dis, devp(m, p + "conv.bias"), nullptr, nullptr, conv))) return rc;
        // edge MLP + scatter-mean + block epilogue in one kernel (k_gn_edge_gate below)
        if ((rc = launch_edge_gate<D, false>(ctx, st, N, row_ptr, eid, csr_dst, edge_attr, devp(m, "#" + p + "edge_inject.proj.0.weightT"),
                                      devp(m, p + "edge_inject.proj.0.bias"), devp(m, "#" + p + "edge_inject.proj.2.weight.p"),
                                      devp(m, p + "edge_inject.proj.2.bias"), conv, bn_of(m, p + "bn."), h, out)))
            return rc;
    }
    for (int s = 0; s < n_states; ++s) {                       // head.0 on the concatenation = sum of per-state products
        GemmArgs a{};
        a.A1 = states + ND * s; a.Wp1 = devp(m, "#head.0.weight.p" + std::to_string(s)); a.out = z0; a.accumulate = s > 0;
        if ((rc = launch_gemm<D, 3>(ctx, st, N, a))) return rc;
    }
    hipLaunchKernelGGL((k_gn_head<D>), dim3(wave_blocks), dim3(256), 0, st, N, z0, devp(m, "head.0.bias"), bn_of(m, "head.1."),
                       devp(m, "#head.4.weightT"), devp(m, "head.4.bias"), devp(m, "head.6.weight"), devp(m, "head.6.bias"),
                       logits, probs);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

} // namespace ggc

extern "C" {

int ggc_gcnnet_configure(ggc_ctx* ctx, int hidden, int n_layers) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, hidden == 32 || hidden == 64 || hidden == 96 || hidden == 128, GGC_E_UNSUPPORTED,
                "hidden_channels=%d unsupported: the MFMA tiling needs a multiple of 32 up to 128", hidden);
    GGC_REQUIRE(ctx, n_layers >= 1 && n_layers <= 30, GGC_E_INVALID_ARG, "n_layers=%d out of range [1,30]", n_layers);
    ResgcnWeights& m = ctx->model2;
    if (m.D != hidden || m.n_layers != n_layers) m.host.clear();
    m.D = hidden; m.n_layers = n_layers; m.dev_ok = false;
    return GGC_OK;
}

int ggc_gcnnet_load_weight(ggc_ctx* ctx, const char* name, const float* data, int64_t numel) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, name && (data || numel == 0) && numel >= 0, GGC_E_INVALID_ARG, "bad weight arguments");
    GGC_REQUIRE(ctx, ctx->model2.D > 0, GGC_E_STATE, "ggc_gcnnet_configure has not been called");
    const std::string key(name), tail = "num_batches_tracked";
    if (key.size() >= tail.size() && key.compare(key.size() - tail.size(), tail.size(), tail) == 0) return GGC_OK;
    bool known = false;
    for (const Need& nd : needed_gcnnet(ctx->model2))
        if (nd.key == key) {
            GGC_REQUIRE(ctx, nd.numel == numel, GGC_E_SHAPE, "weight '%s' has %lld elements, expected %lld", name,
                        (long long)numel, (long long)nd.numel);
            known = true;
            break;
        }
    GGC_REQUIRE(ctx, known, GGC_E_INVALID_ARG, "unexpected state_dict key '%s' for GCNTrimapNet(D=%d, n=%d)", name,
                ctx->model2.D, ctx->model2.n_layers);
    ctx->model2.host[key].assign(data, data + numel);
    ctx->model2.dev_ok = false;
    return GGC_OK;
}

int ggc_gcnnet_ready(ggc_ctx* ctx) {
    if (!ctx) return GGC_E_INVALID_ARG;
    return check_ready_gcnnet(ctx);
}

int ggc_gcnnet_forward(ggc_ctx* ctx, ggc_stream stream, int N, int E, const float* x, const int32_t* edge_src,
                       const int32_t* edge_dst, const float* edge_attr, float* logits, float* probs) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, N >= 1 && E >= 0, GGC_E_SHAPE, "bad sizes N=%d E=%d", N, E);
    GGC_REQUIRE(ctx, x && (E == 0 || (edge_src && edge_dst && edge_attr)), GGC_E_INVALID_ARG, "null input pointer");
    GGC_REQUIRE(ctx, logits || probs, GGC_E_INVALID_ARG, "both outputs are NULL");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    int rc = prepare_weights_gcnnet(ctx);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (ctx->model2.D) {
        case 32:  return forward_gcnnet_t<32>(ctx, st, N, E, x, edge_src, edge_dst, edge_attr, logits, probs);
        case 64:  return forward_gcnnet_t<64>(ctx, st, N, E, x, edge_src, edge_dst, edge_attr, logits, probs);
        case 96:  return forward_gcnnet_t<96>(ctx, st, N, E, x, edge_src, edge_dst, edge_attr, logits, probs);
        case 128: return forward_gcnnet_t<128>(ctx, st, N, E, x, edge_src, edge_dst, edge_attr, logits, probs);
    }
    return set_err(ctx, GGC_E_STATE, "model not configured");
}

} // extern "C"

// ===================================================================================================
// GCNTrimapNet, fused edge gate.  The gate MLP of a block is 316 of the model's 350 GFLOP at batch 256, and done as
// separate passes its E x D intermediates (2 x 824 MB) cross HBM four times per block.  Here a wave owns a contiguous
// range of destination nodes — hence a contiguous range of CSR edge positions — and walks it in tiles of 32 edges:
//   * the first layer relu(W1 e + b1) is generated straight into the MFMA A operand (5 multiply-adds per value);
//   * the D x D layer runs on v_mfma_f32_32x32x2_f32 against W2 packed in LDS, like k_gemm;
//   * sigmoid(. + b2) goes through a small LDS tile and is summed per destination in CSR (= edge) order, and at the
//     end of a destination's edges the block epilogue out = (relu(bn(conv)) + h) * mean is written directly.
// Nothing of size E x D reaches memory.  Same sums in the same order as the unfused kernels.
// ===================================================================================================
namespace ggc {

constexpr int EG_WAVES = 8, EG_NODES = 32;     // waves per block, destination nodes per wave
constexpr int EG_STAGE = 66;                   // row stride of the sigmoid tile: two 32-column tiles + 2 words of padding

template <int D, bool MUL_ONLY>      // MUL_ONLY: out = conv * mean (GATTrimapNet: conv holds gelu(LayerNorm(GATv2)) already)
__global__ void __launch_bounds__(64 * EG_WAVES) k_gn_edge_gate(int N, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ eid,
                                                                const int32_t* __restrict__ csr_dst,
                                                                const float* __restrict__ edge_attr, const float* __restrict__ w1T,
                                                                const float* __restrict__ b1, const float* __restrict__ w2p,
                                                                const float* __restrict__ b2, const float* __restrict__ conv, BnW bn,
                                                                const float* __restrict__ h, float* __restrict__ out) {
    constexpr int T = D / 32, KH = D / 2;
    extern __shared__ float4 eg_smem4[];                       // W2 packed [D*D] | W1T [5][D] | b1 [D] | per wave stage [32][EG_STAGE]
    float* s_w1 = reinterpret_cast<float*>(eg_smem4) + (size_t)D * D;
    float* s_b1 = s_w1 + EDGE_CH * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* stage = s_b1 + D + (size_t)wave * 32 * EG_STAGE;
    const int hk = lane >> 5, li = lane & 31;
    for (int i = tid; i < D * D / 4; i += 64 * EG_WAVES) eg_smem4[i] = reinterpret_cast<const float4*>(w2p)[i];
    for (int i = tid; i < EDGE_CH * D; i += 64 * EG_WAVES) s_w1[i] = w1T[i];
    for (int i = tid; i < D; i += 64 * EG_WAVES) s_b1[i] = b1[i];
    __syncthreads();
    const int n0 = (blockIdx.x * EG_WAVES + wave) * EG_NODES;
    if (n0 >= N) return;                                        // (after the only block barrier)
    const int n1 = min(n0 + EG_NODES, N);
    const int e0 = row_ptr[n0], e1 = row_ptr[n1];

    // epilogue of one destination for this lane's column of column tile t
    auto flush = [&](int node, int t, float sum) {
        const int c = 32 * t + li;
        const int cnt = row_ptr[node + 1] - row_ptr[node];
        const float cf = (float)(cnt > 1 ? cnt : 1);
        float v = conv[(size_t)node * D + c];
        if (!MUL_ONLY) {
            v = bn_apply(v, bn, c);
            v = v > 0.0f ? v : 0.0f;
            v = v + h[(size_t)node * D + c];
        }
        out[(size_t)node * D + c] = v * (sum / cf);
    };
    float sums[T];
#pragma unroll
    for (int t = 0; t < T; ++t) sums[t] = 0.0f;
    int cur = n0;                                               // destination whose edges are being summed (wave-uniform)
    // the tile's inputs (edge attributes through eid, destination of every CSR position) are dependent global loads: the
    // next tile's are fetched while this tile's MFMAs run
    float ea_n[EDGE_CH] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int dnode_n = n1;
    auto fetch = [&](int base) {
        const int j = base + li;
#pragma unroll
        for (int k = 0; k < EDGE_CH; ++k) ea_n[k] = 0.0f;
        dnode_n = n1;
        if (j < e1) {
            const float* a = edge_attr + (size_t)eid[j] * EDGE_CH;
#pragma unroll
            for (int k = 0; k < EDGE_CH; ++k) ea_n[k] = a[k];
            dnode_n = csr_dst[j];
        }
    };
    fetch(e0);
    for (int base = e0; base < e1; base += 32) {
        // ---- A operand: this lane's half row of relu(W1 e + b1) for edge position base + li
        const bool have = base + li < e1;
        float ea[EDGE_CH];
#pragma unroll
        for (int k = 0; k < EDGE_CH; ++k) ea[k] = ea_n[k];
        const int dnode = dnode_n;
        if (base + 32 < e1) fetch(base + 32);
        float a[KH];
#pragma unroll
        for (int s = 0; s < KH; s += 4) {                       // four values at a time on the packed-f32 pipe, same op order
            const int c = hk * KH + s;
            v4f acc4 = 0.0f;
#pragma unroll
            for (int k = 0; k < EDGE_CH; ++k) acc4 += ea[k] * *reinterpret_cast<const v4f*>(s_w1 + k * D + c);
            acc4 += *reinterpret_cast<const v4f*>(s_b1 + c);
            a[s + 0] = (have && acc4.x > 0.0f) ? acc4.x : 0.0f; a[s + 1] = (have && acc4.y > 0.0f) ? acc4.y : 0.0f;
            a[s + 2] = (have && acc4.z > 0.0f) ? acc4.z : 0.0f; a[s + 3] = (have && acc4.w > 0.0f) ? acc4.w : 0.0f;
        }
        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
        for (int s4 = 0; s4 < KH / 4; ++s4) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float4 b = eg_smem4[(s4 * T + t) * 64 + lane];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4 + 0], b.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4 + 1], b.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4 + 2], b.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4 + 3], b.w, acc[t], 0, 0, 0);
            }
        }
        const int n_rows = min(32, e1 - base);
        const int cur_in = cur;
        // ---- two column tiles at a time: sigmoid through the LDS tile, then an ordered walk down the rows in which lane
        // (hk, li) owns column 32 (t + hk) + li
        int c_last = cur_in;
#pragma unroll
        for (int t = 0; t < T; t += 2) {
#pragma unroll
            for (int tt = 0; tt < 2 && t + tt < T; ++tt) {
                const float bias = b2[32 * (t + tt) + li];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // sigmoid as the shared IEEE sequence (include/ggc_fmath.h): the oracle produces the same bits
                    const float z = acc[t + tt][r] + bias;
                    stage[((r & 3) + 8 * (r >> 2) + 4 * hk) * EG_STAGE + 32 * tt + li] = ggc_sigmoid_nr(z);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int mt = t + hk;                              // this lane's column tile in the walk
            const bool mine = mt < T;
            float sum = 0.0f;
#pragma unroll
            for (int q = 0; q < T; ++q) if (q == mt) sum = sums[q];
            int c_node = cur_in;
            for (int row = 0; row < n_rows; ++row) {
                const int nd = __shfl(dnode, row, 64);          // wave-uniform
                if (nd != c_node) {
                    if (mine) {
                        flush(c_node, mt, sum);
                        for (int z = c_node + 1; z < nd; ++z) flush(z, mt, 0.0f);     // destinations without edges
                    }
                    c_node = nd; sum = 0.0f;
                }
                sum += stage[row * EG_STAGE + 32 * hk + li];
            }
#pragma unroll
            for (int q = 0; q < T; ++q) if (q == mt) sums[q] = sum;
            c_last = c_node;
            __builtin_amdgcn_wave_barrier();
        }
        cur = c_last;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if ((t & 1) == hk) {                                    // the lane half that summed this column tile
            flush(cur, t, sums[t]);
            for (int z = cur + 1; z < n1; ++z) flush(z, t, 0.0f);
        }
    }
}

template <int D, bool MUL_ONLY>
static int launch_edge_gate(ggc_ctx* ctx, hipStream_t st, int N, const int32_t* row_ptr, const int32_t* eid, const int32_t* csr_dst,
                            const float* edge_attr, const float* w1T, const float* b1, const float* w2p, const float* b2,
                            const float* conv, const BnW& bn, const float* h, float* out) {
    const size_t lds = ((size_t)D * D + (size_t)EDGE_CH * D + D + (size_t)EG_WAVES * 32 * EG_STAGE) * sizeof(float);
    static DeviceOnce attr_set;
    if (attr_set.need(ctx->device)) {
        GGC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gn_edge_gate<D, MUL_ONLY>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
        attr_set.done(ctx->device);
    }
    ProfScope prof(ctx, st, "gcnnet_edge_gate");
    hipLaunchKernelGGL((k_gn_edge_gate<D, MUL_ONLY>), dim3(cdiv(N, EG_WAVES * EG_NODES)), dim3(64 * EG_WAVES), lds, st, N, row_ptr, eid, csr_dst,
                       edge_attr, w1T, b1, w2p, b2, conv, bn, h, out);
    GGC_LAUNCH_CHECK(ctx);
    return GGC_OK;
}

} // namespace ggc

// ===================================================================================================
// GATTrimapNet (reference model.py:323-414; SURVEY.md section 8(f), last rank): GATv2 attention with edge features.
//   h0 = GELU(LN(Linear(BN(x))));  skip = skip_proj(h0)
//   5 x { GATv2Conv(h) -> LN -> GELU -> EdgeInjectionLayer }   ;   h + skip -> GlobalContextModule -> head
// GATv2Conv (PyG 2.x semantics, restated from its documentation — the library is absent, parity with it unpinned):
// x_l = lin_l(x), x_r = lin_r(x) (both with bias), one self loop per node whose edge attribute is the MEAN of the node's
// incoming edge attributes (fill_value="mean"); for an edge j -> i and head h
//     m = leaky_relu(x_r[i] + x_l[j] + lin_edge(e_ij), 0.2);   a = att[h] . m[h];   alpha = softmax over the edges into i
//     out_i[h] = sum_j alpha_ij x_l[j][h];   concat heads, + bias.
// One wave per destination node (lane l holds channels l, l + 64): the per-head dot product is a butterfly over the head's
// C = D / heads consecutive lanes; the softmax is two passes over the node's incoming edges in CSR (= edge) order, the
// self loop last.  LayerNorm + GELU of the block are fused in (the wave holds the whole output row).  The per-block edge
// gate is the fused MFMA kernel of GCNTrimapNet with a multiply-only epilogue; the D x D products run on k_gemm.
// ===================================================================================================
namespace ggc {

template <int D>
__global__ void __launch_bounds__(256) k_gat_input(int N, const float* __restrict__ x, BnW bn_in, const float* __restrict__ w_inT,
                                                   const float* __restrict__ b_in, const float* __restrict__ ln_w,
                                                   const float* __restrict__ ln_b, float* __restrict__ h) {
    constexpr int NC = (D + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int node = wave; node < N; node += n_waves) {
        float xn[IN_CH];
#pragma unroll
        for (int k = 0; k < IN_CH; ++k) xn[k] = bn_apply(x[(size_t)node * IN_CH + k], bn_in, k);
        float a[NC];
        float s1 = 0.0f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            float acc = 0.0f;
            if (c < D) {
#pragma unroll
                for (int k = 0; k < IN_CH; ++k) acc += xn[k] * w_inT[k * D + c];
                acc += b_in[c];
                s1 += acc;
            }
            a[j] = acc;
        }
        const float mean = wave_sum(s1) / (float)D;
        float s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NC; ++j) { const int c = lane + 64 * j; if (c < D) { const float dv = a[j] - mean; s2 += dv * dv; } }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)D + 1e-5f);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            if (c < D) h[(size_t)node * D + c] = gelu_f((a[j] - mean) * rstd * ln_w[c] + ln_b[c]);
        }
    }
}

struct GatW { const float *bl, *br, *weT /*[5][D]*/, *att /*[D]*/, *bias, *ln_w, *ln_b; };

template <int D, int HEADS>
__global__ void __launch_bounds__(256) k_gat_attn(int N, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                  const int32_t* __restrict__ eid, const float* __restrict__ edge_attr,
                                                  const float* __restrict__ xl, const float* __restrict__ xr, GatW w,
                                                  float* __restrict__ out) {
    constexpr int NC = (D + 63) / 64, C = D / HEADS;          // a head is C consecutive channels: C <= 64 consecutive lanes of one
                                                              // register, or (C = 128: D = 128, one head) all lanes of both
    static_assert(C >= 4 && (C & (C - 1)) == 0 && (C <= 64 || (C == 128 && NC == 2)), "head width: a power of two from 4 to 128");
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    float bl[NC], att[NC], we[NC][EDGE_CH];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = lane + 64 * j;
        bl[j] = c < D ? w.bl[c] : 0.0f; att[j] = c < D ? w.att[c] : 0.0f;
#pragma unroll
        for (int k = 0; k < EDGE_CH; ++k) we[j][k] = c < D ? w.weT[k * D + c] : 0.0f;
    }
    for (int node = wave; node < N; node += n_waves) {
        const int beg = row_ptr[node], end = row_ptr[node + 1], cnt = end - beg;
        float xri[NC], xli[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            xri[j] = c < D ? xr[(size_t)node * D + c] + w.br[c] : 0.0f;
            xli[j] = c < D ? xl[(size_t)node * D + c] + bl[j] : 0.0f;
        }
        // the self loop's edge attribute: mean of the incoming ones (sum in edge order / count; zeros without edges)
        float am[EDGE_CH] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = beg; p < end; ++p) {
            const float* a = edge_attr + (size_t)eid[p] * EDGE_CH;
#pragma unroll
            for (int k = 0; k < EDGE_CH; ++k) am[k] += a[k];
        }
        const float cf = (float)(cnt > 0 ? cnt : 1);
#pragma unroll
        for (int k = 0; k < EDGE_CH; ++k) am[k] = am[k] / cf;
        // attention logit of one edge for this lane's head(s): every lane of a head ends with the head's value
        auto logit = [&](const float* a, const float (&xlj)[NC], float (&lg)[NC]) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                float ev = 0.0f;
#pragma unroll
                for (int k = 0; k < EDGE_CH; ++k) ev += a[k] * we[j][k];
                float m = (xri[j] + xlj[j]) + ev;
                m = m > 0.0f ? m : 0.2f * m;
                float v = m * att[j];
#pragma unroll
                for (int o = (C < 64 ? C : 64) / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                lg[j] = v;
            }
            if (C == 128) { const float t = lg[0] + lg[NC - 1]; lg[0] = t; lg[NC - 1] = t; }     // one head across both registers: low half + high half
        };
        float mx[NC], lgs[NC];
        logit(am, xli, lgs);                                   // self loop
#pragma unroll
        for (int j = 0; j < NC; ++j) mx[j] = lgs[j];
        for (int p = beg; p < end; ++p) {
            const int src = col[p];
            float xlj[NC], lg[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) { const int c = lane + 64 * j; xlj[j] = c < D ? xl[(size_t)src * D + c] + bl[j] : 0.0f; }
            logit(edge_attr + (size_t)eid[p] * EDGE_CH, xlj, lg);
#pragma unroll
            for (int j = 0; j < NC; ++j) mx[j] = fmaxf(mx[j], lg[j]);
        }
        float ssum[NC], acc[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) { ssum[j] = 0.0f; acc[j] = 0.0f; }
        for (int p = beg; p < end; ++p) {                      // the edges in order ...
            const int src = col[p];
            float xlj[NC], lg[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) { const int c = lane + 64 * j; xlj[j] = c < D ? xl[(size_t)src * D + c] + bl[j] : 0.0f; }
            logit(edge_attr + (size_t)eid[p] * EDGE_CH, xlj, lg);
#pragma unroll
            for (int j = 0; j < NC; ++j) { const float e = ggc_expf(lg[j] - mx[j]); ssum[j] += e; acc[j] += e * xlj[j]; }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) { const float e = ggc_expf(lgs[j] - mx[j]); ssum[j] += e; acc[j] += e * xli[j]; }    // ... the self loop last
        // + bias, LayerNorm, GELU
        float o[NC];
        float s1 = 0.0f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            o[j] = c < D ? acc[j] / (ssum[j] + 1e-16f) + w.bias[c] : 0.0f;
            if (c < D) s1 += o[j];
        }
        const float mean = wave_sum(s1) / (float)D;
        float s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NC; ++j) { const int c = lane + 64 * j; if (c < D) { const float dv = o[j] - mean; s2 += dv * dv; } }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)D + 1e-5f);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            if (c < D) out[(size_t)node * D + c] = gelu_f((o[j] - mean) * rstd * w.ln_w[c] + w.ln_b[c]);
        }
    }
}

// h + skip and the readout score attn . (h + skip) + b
template <int D>
__global__ void __launch_bounds__(256) k_gat_score(int N, const float* __restrict__ h, const float* __restrict__ skip,
                                                   const float* __restrict__ attn_w, const float* __restrict__ attn_b,
                                                   float* __restrict__ hs, float* __restrict__ score) {
    constexpr int NC = (D + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int node = wave; node < N; node += n_waves) {
        float dot = 0.0f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                const float v = h[(size_t)node * D + c] + skip[(size_t)node * D + c];
                hs[(size_t)node * D + c] = v;
                dot += v * attn_w[c];
            }
        }
        dot = wave_sum(dot);
        if (lane == 0) score[node] = dot + attn_b[0];
    }
}

static std::vector<Need> needed_gat(const ResgcnWeights& m) {
    const int D = m.D, n = m.n_layers;
    std::vector<Need> v;
    for (const char* k : BN_KEYS) v.push_back({std::string("in_norm.norm.") + k, IN_CH});
    v.push_back({"input_proj.0.weight", (int64_t)D * IN_CH}); v.push_back({"input_proj.0.bias", D});
    v.push_back({"input_proj.1.weight", D}); v.push_back({"input_proj.1.bias", D});
    for (int i = 0; i < n; ++i) {
        const std::string s = std::to_string(i);
        v.push_back({"convs." + s + ".att", D});
        v.push_back({"convs." + s + ".lin_l.weight", (int64_t)D * D}); v.push_back({"convs." + s + ".lin_l.bias", D});
        v.push_back({"convs." + s + ".lin_r.weight", (int64_t)D * D}); v.push_back({"convs." + s + ".lin_r.bias", D});
        v.push_back({"convs." + s + ".lin_edge.weight", (int64_t)D * EDGE_CH}); v.push_back({"convs." + s + ".bias", D});
        v.push_back({"lns." + s + ".weight", D}); v.push_back({"lns." + s + ".bias", D});
        v.push_back({"edge_gates." + s + ".proj.0.weight", (int64_t)D * EDGE_CH}); v.push_back({"edge_gates." + s + ".proj.0.bias", D});
        v.push_back({"edge_gates." + s + ".proj.2.weight", (int64_t)D * D}); v.push_back({"edge_gates." + s + ".proj.2.bias", D});
    }
    v.push_back({"skip_proj.weight", (int64_t)D * D});
    v.push_back({"ctx.attn.weight", D}); v.push_back({"ctx.attn.bias", 1});
    v.push_back({"ctx.compress.weight", (int64_t)(D / 2) * D}); v.push_back({"ctx.compress.bias", D / 2});
    v.push_back({"ctx.expand.weight", (int64_t)D * (D / 2)}); v.push_back({"ctx.expand.bias", D});
    v.push_back({"head.0.weight", (int64_t)D * D}); v.push_back({"head.0.bias", D});
    v.push_back({"head.3.weight", (int64_t)N_CLS * D}); v.push_back({"head.3.bias", N_CLS});
    return v;
}

static int check_ready_gat(ggc_ctx* ctx) {
    ResgcnWeights& m = ctx->model3;
    GGC_REQUIRE(ctx, m.D > 0, GGC_E_STATE, "ggc_gat_configure has not been called");
    for (const Need& nd : needed_gat(m)) {
        auto it = m.host.find(nd.key);
        GGC_REQUIRE(ctx, it != m.host.end(), GGC_E_STATE, "missing weight '%s'", nd.key.c_str());
        GGC_REQUIRE(ctx, (int64_t)it->second.size() == nd.numel, GGC_E_SHAPE, "weight '%s' has %zu elements, expected %lld",
                    nd.key.c_str(), it->second.size(), (long long)nd.numel);
    }
    return GGC_OK;
}

static int prepare_weights_gat(ggc_ctx* ctx) {
    ResgcnWeights& m = ctx->model3;
    if (m.dev_ok) return GGC_OK;
    int rc = check_ready_gat(ctx);
    if (rc) return rc;
    GGC_HIP(ctx, hipDeviceSynchronize());                  // (see prepare_weights)
    const int D = m.D, n = m.n_layers;
    for (auto& kv : m.host) { if ((rc = upload(ctx, m, kv.first, kv.second))) return rc; }
    if ((rc = upload(ctx, m, "#input_proj.0.weightT", transpose(m.host["input_proj.0.weight"], D, IN_CH)))) return rc;
    if ((rc = upload(ctx, m, "#skip_proj.weight.p", pack_mfma(m.host["skip_proj.weight"], D)))) return rc;
    if ((rc = upload(ctx, m, "#ctx.compress.weightT", transpose(m.host["ctx.compress.weight"], D / 2, D)))) return rc;
    if ((rc = upload(ctx, m, "#ctx.expand.weightT", transpose(m.host["ctx.expand.weight"], D, D / 2)))) return rc;
    if ((rc = upload(ctx, m, "#head.0.weight.p", pack_mfma(m.host["head.0.weight"], D)))) return rc;
    for (int i = 0; i < n; ++i) {
        const std::string c = "convs." + std::to_string(i) + ".", g = "edge_gates." + std::to_string(i) + ".";
        if ((rc = upload(ctx, m, "#" + c + "lin_l.weight.p", pack_mfma(m.host[c + "lin_l.weight"], D)))) return rc;
        if ((rc = upload(ctx, m, "#" + c + "lin_r.weight.p", pack_mfma(m.host[c + "lin_r.weight"], D)))) return rc;
        if ((rc = upload(ctx, m, "#" + c + "lin_edge.weightT", transpose(m.host[c + "lin_edge.weight"], D, EDGE_CH)))) return rc;
        if ((rc = upload(ctx, m, "#" + g + "proj.0.weightT", transpose(m.host[g + "proj.0.weight"], D, EDGE_CH)))) return rc;
        if ((rc = upload(ctx, m, "#" + g + "proj.2.weight.p", pack_mfma(m.host[g + "proj.2.weight"], D)))) return rc;
    }
    m.dev_ok = true;
    return GGC_OK;
}

template <int D>
static int forward_gat_t(ggc_ctx* ctx, hipStream_t st, int G, int N, int E, const float* x, const int32_t* edge_src,
                         const int32_t* edge_dst, const float* edge_attr, const int32_t* node_ptr, float* logits, float* probs) {
    ResgcnWeights& m = ctx->model3;
    const int heads = m.Q;
    const int n = m.n_layers;
    const size_t ND = (size_t)N * D;
    int32_t* row_ptr = scratch_t<int32_t>(ctx, S_CSR_ROWPTR, (size_t)N + 1);
    int32_t* col = scratch_t<int32_t>(ctx, S_CSR_COL, (size_t)std::max(E, 1));
    int32_t* eid = scratch_t<int32_t>(ctx, S_CSR_EID, (size_t)std::max(E, 1));
    int32_t* cursor = scratch_t<int32_t>(ctx, S_CSR_CURSOR, (size_t)N + 1);
    float* dis = scratch_t<float>(ctx, S_DIS, (size_t)N);
    int32_t* batch = scratch_t<int32_t>(ctx, S_BATCH, (size_t)N);
    float* buf = scratch_t<float>(ctx, S_STATES, ND * 4);        // h (ping) | h (pong) | skip | gelu(LN(conv))
    float* xl = scratch_t<float>(ctx, S_XW, ND);
    float* xr = scratch_t<float>(ctx, S_AGG, ND);
    float* hs = scratch_t<float>(ctx, S_HJK, ND);
    float* score = scratch_t<float>(ctx, S_SCORE, (size_t)N);
    float* gvec = scratch_t<float>(ctx, S_GVEC, (size_t)G * D);
    if (!row_ptr || !col || !eid || !cursor || !dis || !batch || !buf || !xl || !xr || !hs || !score || !gvec) return GGC_E_OOM;
    int rc = build_csr(ctx, st, N, E, edge_src, edge_dst, row_ptr, col, eid, cursor, dis);
    if (rc) return rc;
    int32_t* csr_dst = scratch_t<int32_t>(ctx, S_AGG_PACK, (size_t)std::max(E, 1));
    if (!csr_dst) return GGC_E_OOM;
    if (E > 0) {
        hipLaunchKernelGGL(k_csr_dst, dim3(cdiv(E, 256)), dim3(256), 0, st, N, E, row_ptr, csr_dst);
        GGC_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(k_fill_batch, dim3(min(cdiv(N, 256), 4096)), dim3(256), 0, st, G, N, node_ptr, batch);
    GGC_LAUNCH_CHECK(ctx);
    const int wave_blocks = min(cdiv(N, 4), 8 * ctx->n_cu);
    float *h = buf, *h2 = buf + ND, *skip = buf + 2 * ND, *act = buf + 3 * ND;
    hipLaunchKernelGGL((k_gat_input<D>), dim3(wave_blocks), dim3(256), 0, st, N, x, bn_of(m, "in_norm.norm."),
                       devp(m, "#input_proj.0.weightT"), devp(m, "input_proj.0.bias"), devp(m, "input_proj.1.weight"),
                       devp(m, "input_proj.1.bias"), h);
    GGC_LAUNCH_CHECK(ctx);
    {
        GemmArgs a{};
        a.A1 = h; a.Wp1 = devp(m, "#skip_proj.weight.p"); a.out = skip;
        if ((rc = launch_gemm<D, 3>(ctx, st, N, a))) return rc;
    }
    for (int l = 0; l < n; ++l) {
        const std::string c = "convs." + std::to_string(l) + ".", g = "edge_gates." + std::to_string(l) + ".", ln = "lns." + std::to_string(l) + ".";
        GemmArgs a{};
        a.A1 = h; a.Wp1 = devp(m, "#" + c + "lin_l.weight.p"); a.out = xl;
        if ((rc = launch_gemm<D, 3>(ctx, st, N, a))) return rc;
        a.Wp1 = devp(m, "#" + c + "lin_r.weight.p"); a.out = xr;
        if ((rc = launch_gemm<D, 3>(ctx, st, N, a))) return rc;
        GatW w{devp(m, c + "lin_l.bias"), devp(m, c + "lin_r.bias"), devp(m, "#" + c + "lin_edge.weightT"), devp(m, c + "att"),
               devp(m, c + "bias"), devp(m, ln + "weight"), devp(m, ln + "bias")};
        {
            ProfScope prof(ctx, st, "gat_attention");
            if (heads == 8) hipLaunchKernelGGL((k_gat_attn<D, 8>), dim3(wave_blocks), dim3(256), 0, st, N, row_ptr, col, eid, edge_attr, xl, xr, w, act);
            else if (heads == 4) hipLaunchKernelGGL((k_gat_attn<D, 4>), dim3(wave_blocks), dim3(256), 0, st, N, row_ptr, col, eid, edge_attr, xl, xr, w, act);
            else if (heads == 2) hipLaunchKernelGGL((k_gat_attn<D, 2>), dim3(wave_blocks), dim3(256), 0, st, N, row_ptr, col, eid, edge_attr, xl, xr, w, act);
            else hipLaunchKernelGGL((k_gat_attn<D, 1>), dim3(wave_blocks), dim3(256), 0, st, N, row_ptr, col, eid, edge_attr, xl, xr, w, act);
        }
        GGC_LAUNCH_CHECK(ctx);
        if ((rc = launch_edge_gate<D, true>(ctx, st, N, row_ptr, eid, csr_dst, edge_attr, devp(m, "#" + g + "proj.0.weightT"),
                                            devp(m, g + "proj.0.bias"), devp(m, "#" + g + "proj.2.weight.p"), devp(m, g + "proj.2.bias"),
                                            act, BnW{}, nullptr, h2)))
            return rc;
        std::swap(h, h2);
    }
    hipLaunchKernelGGL((k_gat_score<D>), dim3(wave_blocks), dim3(256), 0, st, N, h, skip, devp(m, "ctx.attn.weight"),
                       devp(m, "ctx.attn.bias"), hs, score);
    GGC_LAUNCH_CHECK(ctx);
    {
        CtxW w{devp(m, "#ctx.compress.weightT"), devp(m, "ctx.compress.bias"), devp(m, "#ctx.expand.weightT"), devp(m, "ctx.expand.bias")};
        hipLaunchKernelGGL((k_graph_ctx<D>), dim3(G), dim3(256), 0, st, node_ptr, score, hs, w, gvec);
        GGC_LAUNCH_CHECK(ctx);
    }
    {
        GemmArgs a{};
        a.A1 = hs; a.Wp1 = devp(m, "#head.0.weight.p"); a.bias = devp(m, "head.0.bias");
        a.batch = batch; a.gvec = gvec;
        a.ep_w = devp(m, "head.3.weight"); a.ep_b = devp(m, "head.3.bias");
        a.out = logits; a.out2 = probs;
        if ((rc = launch_gemm<D, 4>(ctx, st, N, a))) return rc;
    }
    return GGC_OK;
}

} // namespace ggc

extern "C" {

int ggc_gat_configure(ggc_ctx* ctx, int hidden, int n_heads, int n_layers) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, hidden == 32 || hidden == 64 || hidden == 128, GGC_E_UNSUPPORTED,
                "hidden_channels=%d unsupported: GATTrimapNet runs at 32, 64 or 128 (a head must span a power-of-two number of lanes)", hidden);
    GGC_REQUIRE(ctx, n_heads == 1 || n_heads == 2 || n_heads == 4 || n_heads == 8, GGC_E_UNSUPPORTED,
                "n_heads=%d unsupported: 1, 2, 4 or 8 (the reference's default is 8)", n_heads);
    GGC_REQUIRE(ctx, n_layers >= 1 && n_layers <= 30, GGC_E_INVALID_ARG, "n_layers=%d out of range [1,30]", n_layers);
    ResgcnWeights& m = ctx->model3;
    if (m.D != hidden || m.n_layers != n_layers) m.host.clear();     // (the head count changes no weight shape: att is [1, H, D / H] = D values)
    m.D = hidden; m.n_layers = n_layers; m.Q = n_heads; m.dev_ok = false;
    return GGC_OK;
}

int ggc_gat_load_weight(ggc_ctx* ctx, const char* name, const float* data, int64_t numel) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, name && (data || numel == 0) && numel >= 0, GGC_E_INVALID_ARG, "bad weight arguments");
    GGC_REQUIRE(ctx, ctx->model3.D > 0, GGC_E_STATE, "ggc_gat_configure has not been called");
    const std::string key(name), tail = "num_batches_tracked";
    if (key.size() >= tail.size() && key.compare(key.size() - tail.size(), tail.size(), tail) == 0) return GGC_OK;
    bool known = false;
    for (const Need& nd : needed_gat(ctx->model3))
        if (nd.key == key) {
            GGC_REQUIRE(ctx, nd.numel == numel, GGC_E_SHAPE, "weight '%s' has %lld elements, expected %lld", name,
                        (long long)numel, (long long)nd.numel);
            known = true;
            break;
        }
    GGC_REQUIRE(ctx, known, GGC_E_INVALID_ARG, "unexpected state_dict key '%s' for GATTrimapNet(D=%d, n=%d)", name,
                ctx->model3.D, ctx->model3.n_layers);
    ctx->model3.host[key].assign(data, data + numel);
    ctx->model3.dev_ok = false;
    return GGC_OK;
}

int ggc_gat_ready(ggc_ctx* ctx) {
    if (!ctx) return GGC_E_INVALID_ARG;
    return check_ready_gat(ctx);
}

int ggc_gat_forward(ggc_ctx* ctx, ggc_stream stream, int G, int N, int E, const float* x, const int32_t* edge_src,
                    const int32_t* edge_dst, const float* edge_attr, const int32_t* node_ptr, float* logits, float* probs) {
    if (!ctx) return GGC_E_INVALID_ARG;
    GGC_REQUIRE(ctx, G >= 1 && N >= 1 && E >= 0, GGC_E_SHAPE, "bad sizes G=%d N=%d E=%d", G, N, E);
    GGC_REQUIRE(ctx, x && node_ptr && (E == 0 || (edge_src && edge_dst && edge_attr)), GGC_E_INVALID_ARG, "null input pointer");
    GGC_REQUIRE(ctx, logits || probs, GGC_E_INVALID_ARG, "both outputs are NULL");
    GGC_HIP(ctx, hipSetDevice(ctx->device));
    int rc = prepare_weights_gat(ctx);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (ctx->model3.D) {
        case 32:  return forward_gat_t<32>(ctx, st, G, N, E, x, edge_src, edge_dst, edge_attr, node_ptr, logits, probs);
        case 64:  return forward_gat_t<64>(ctx, st, G, N, E, x, edge_src, edge_dst, edge_attr, node_ptr, logits, probs);
        case 128: return forward_gat_t<128>(ctx, st, G, N, E, x, edge_src, edge_dst, edge_attr, node_ptr, logits, probs);
    }
    return set_err(ctx, GGC_E_STATE, "model not configured");
}

} // extern "C"
