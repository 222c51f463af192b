// Micro-test 2 for the pooled max-flow: how do device-scope atomics and sc1 loads mix?
//  P1  CU A: atomicAdd(word, 1) (no return), drained, then flag via atomicExch.  CU B: waits on the flag with atomic reads,
//      then reads the word with an sc1 LOAD.  Stale if the atomic did not land where sc1 loads are served from.
//  P2  CU A: atomicExch(word, r).  CU B polls the word with sc1 LOADS only (bounded).  Never seen = stale forever.
// Each for pairs on the same XCD (b, b+8) and on different XCDs (b, b+1); placement verified with HW_REG_XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }
__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(64) k_p1(int stride, int* words, int* flags, int rounds, int* bad, int* timeout, int* same_xcc) {
    const int b = blockIdx.x, grp = b / (2 * stride), off = b % (2 * stride);
    const bool is_a = off < stride;
    const int p = is_a ? b + stride : b - stride;
    int* w = words + (size_t)(is_a ? b : p) * 64;          // 64 words per pair, one per lane
    int* f = flags + (is_a ? b : p) * 32;
    if (threadIdx.x == 0 && !is_a) { /* record whether the pair shares an XCC: A writes its id first */ }
    __shared__ int s_x;
    if (is_a) { if (threadIdx.x == 0) atomicExch(&f[1], xcc_id() + 1); }
    else if (threadIdx.x == 0) { int x; long long sp = 0; while ((x = atomicAdd(&f[1], 0)) == 0 && ++sp < (1 << 22)) {} atomicAdd(&same_xcc[(x - 1) == xcc_id()], 1); }
    int nbad = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (is_a) {
            atomicAdd(&w[threadIdx.x], 1);                                     // no return value used
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) atomicExch(&f[0], r);
            // wait for B's ack so that rounds do not overlap
            if (threadIdx.x == 0) { long long sp = 0; while (atomicAdd(&f[2], 0) < r) { if (++sp > (1ll << 22)) { atomicExch(timeout, 1); break; } } }
        } else {
            if (threadIdx.x == 0) { long long sp = 0; while (atomicAdd(&f[0], 0) < r) { if (++sp > (1ll << 22)) { atomicExch(timeout, 1); break; } } }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            const int v = ld_sc1(&w[threadIdx.x]);
            nbad += (v != r);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) atomicExch(&f[2], r);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ void __launch_bounds__(64) k_p2(int stride, int* words, int rounds, int* never, int* polls) {
    const int b = blockIdx.x, off = b % (2 * stride);
    const bool is_a = off < stride;
    const int p = is_a ? b + stride : b - stride;
    int* w = words + (size_t)(is_a ? b : p) * 64;
    if (threadIdx.x != 0) return;
    for (int r = 1; r <= rounds; ++r) {
        if (is_a) {
            atomicExch(&w[0], r);
            long long sp = 0; while (atomicAdd(&w[1], 0) < r) { if (++sp > (1ll << 20)) break; }   // ack (atomic)
        } else {
            long long sp = 0; bool seen = false;
            while (sp < (1 << 16)) { ++sp; if (ld_sc1(&w[0]) >= r) { seen = true; break; } }
            if (!seen) atomicAdd(never, 1);
            atomicAdd(polls, (int)(sp > 1000000 ? 1000000 : sp));
            atomicExch(&w[1], r);
        }
    }
}

int main() {
    const int N = 256;
    int *words, *flags, *bad, *timeout, *same;
    CK(hipMalloc(&words, (size_t)N * 256)); CK(hipMalloc(&flags, N * 128)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&timeout, 4)); CK(hipMalloc(&same, 8));
    for (int stride : {8, 1}) {
        CK(hipMemset(words, 0, (size_t)N * 256)); CK(hipMemset(flags, 0, N * 128)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(timeout, 0, 4)); CK(hipMemset(same, 0, 8));
        hipLaunchKernelGGL(k_p1, dim3(N), dim3(64), 0, 0, stride, words, flags, 20000, bad, timeout, same);
        CK(hipDeviceSynchronize());
        int hb, ht, hs[2]; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht, timeout, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, same, 8, hipMemcpyDeviceToHost));
        printf("P1 atomicAdd then sc1 load, pairs (b, b+%d): %d pairs share an XCC, %d do not; stale %d of %lld, timeout %d\n", stride, hs[1], hs[0], hb, 20000ll * 64 * N / 2, ht);
    }
    int *never, *polls; CK(hipMalloc(&never, 4)); CK(hipMalloc(&polls, 4));
    for (int stride : {8, 1}) {
        CK(hipMemset(words, 0, (size_t)N * 256)); CK(hipMemset(never, 0, 4)); CK(hipMemset(polls, 0, 4));
        hipLaunchKernelGGL(k_p2, dim3(N), dim3(64), 0, 0, stride, words, 2000, never, polls);
        CK(hipDeviceSynchronize());
        int hn, hp; CK(hipMemcpy(&hn, never, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hp, polls, 4, hipMemcpyDeviceToHost));
        printf("P2 atomicExch then sc1-load polling, pairs (b, b+%d): never seen %d of %d, mean polls %.1f\n", stride, hn, 2000 * N / 2, (double)hp / (2000.0 * N / 2));
    }
    return 0;
}
