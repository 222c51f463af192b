// Micro-test for the pooled max-flow design: (1) HW_REG_XCC_ID per workgroup, (2) is a plain store, drained with
// s_waitcnt vmcnt(0) and announced by a device-scope atomic, visible to an sc1 load on ANOTHER CU of the SAME XCD?
// Producer WG p writes a 4 KB slab with value `round`, drains, bumps flag[p]; consumer WG reads flag (atomic), then the
// slab with sc1 loads and counts mismatches.  Pairs are formed among WGs that report the same XCC id (and, as the
// control experiment, different XCC ids).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }
__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void k_ids(int* xcc, int* cu) {
    if (threadIdx.x == 0) { xcc[blockIdx.x] = xcc_id(); cu[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4) | (__builtin_amdgcn_s_getreg((1 << 11) | (13 << 6) | 4) << 4); }
}

// role[b]: partner block id; even "slot" = producer first.  rounds ping-pong: A writes slabA(round), B checks it and
// writes slabB(round), A checks it, ...
__global__ void __launch_bounds__(256) k_pingpong(const int* partner, const int* is_a, int* slabs, int* flags, int rounds, int* bad, int* timeout, int plain_loads) {
    const int b = blockIdx.x, p = partner[b];
    if (p < 0) return;
    int* mine = slabs + (size_t)b * 1024;
    const int* theirs = slabs + (size_t)p * 1024;
    int nbad = 0;
    for (int r = 1; r <= rounds; ++r) {
        const bool my_turn_first = is_a[b];
        for (int half = 0; half < 2; ++half) {
            const bool produce = (half == 0) == my_turn_first;
            if (produce) {
                for (int i = threadIdx.x; i < 1024; i += 256) mine[i] = r;           // plain stores
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) atomicExch(&flags[b], r);                       // device-scope atomic
            } else {
                if (threadIdx.x == 0) {
                    long long spins = 0;
                    while (atomicAdd(&flags[p], 0) < r) { if (++spins > (1ll << 24)) { atomicExch(timeout, 1); break; } __builtin_amdgcn_s_sleep(2); }
                }
                __syncthreads();
                for (int i = threadIdx.x; i < 1024; i += 256) {
                    const int v = plain_loads ? theirs[i] : ld_sc1(theirs + i);
                    nbad += (v != r);
                }
            }
            __syncthreads();
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const int G = 512;
    int *xcc, *cu; CK(hipMalloc(&xcc, G * 4)); CK(hipMalloc(&cu, G * 4));
    hipLaunchKernelGGL(k_ids, dim3(G), dim3(256), 0, 0, xcc, cu);
    std::vector<int> hx(G), hc(G);
    CK(hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), cu, G * 4, hipMemcpyDeviceToHost));
    int cnt[16] = {};
    int rr_ok = 0;
    for (int b = 0; b < G; ++b) { cnt[hx[b] & 15]++; rr_ok += (hx[b] == hx[b % 8]); }
    printf("XCC ids of %d workgroups:", G); for (int i = 0; i < 16; ++i) if (cnt[i]) printf(" [%d]=%d", i, cnt[i]);
    printf("\nblocks b and b%%8 share an XCC: %d of %d; first 16 ids:", rr_ok, G); for (int b = 0; b < 16; ++b) printf(" %d", hx[b]); printf("\n");
    // resident grid of 256 WGs (1 per CU): pair (b, b+8) -> same XCD if round robin, (b, b+1) -> different
    const int N = 256;
    int *partner, *is_a, *slabs, *flags, *bad, *timeout;
    CK(hipMalloc(&partner, N * 4)); CK(hipMalloc(&is_a, N * 4)); CK(hipMalloc(&slabs, (size_t)N * 4096)); CK(hipMalloc(&flags, N * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&timeout, 4));
    for (int mode = 0; mode < 4; ++mode) {
        const int stride = (mode & 1) ? 1 : 8, plain = (mode >> 1) & 1;
        std::vector<int> hp(N, -1), ha(N, 0);
        for (int b = 0; b < N; ++b) {
            const int grp = b / (2 * stride), off = b % (2 * stride);
            const int q = off < stride ? b + stride : b - stride;
            if (q < N) { hp[b] = q; ha[b] = off < stride; }
            (void)grp;
        }
        CK(hipMemcpy(partner, hp.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(is_a, ha.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(slabs, 0, (size_t)N * 4096)); CK(hipMemset(flags, 0, N * 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(timeout, 0, 4));
        hipLaunchKernelGGL(k_pingpong, dim3(N), dim3(256), 0, 0, partner, is_a, slabs, flags, 2000, bad, timeout, plain);
        CK(hipDeviceSynchronize());
        int hb = 0, ht = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht, timeout, 4, hipMemcpyDeviceToHost));
        // which pairs really shared an XCC in THIS launch is not known (ids are per launch); report by construction
        printf("pairs (b, b+%d) [%s under round-robin], %s loads: %d stale values of %lld read, timeout=%d\n", stride,
               stride == 8 ? "same XCD" : "different XCDs", plain ? "plain" : "sc1", hb, 2000ll * 1024 * N, ht);
    }
    return 0;
}
