// Micro-test: round-trip latency of one lane's dependent memory operations on MI355X, by scope.
//   A  agent-scope returning atomicAdd            (what the asynchronous max-flow kernels use across XCDs)
//   W  workgroup-scope returning atomicAdd        (executes in the XCD's own L2?)
//   L  sc1 (agent-scope relaxed) load             L0 plain load of a line this CU has read before (L1 hit)
//   S  plain store + s_waitcnt vmcnt(0)
// idle chip (1 block) and with 1023 other blocks streaming.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_lat(int* buf, long long* out, int n, float* stream, size_t stream_n) {
    if (blockIdx.x != 0) {                                   // background load
        float acc = 0.f;
        for (int rep = 0; rep < 64; ++rep)
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < stream_n; i += (size_t)gridDim.x * 256) acc += stream[i];
        if (acc == 123.456f) stream[0] = acc;
        return;
    }
    if (threadIdx.x != 0) return;
    int* p = buf + 4096;
    long long t0, t1; int v = 0;
    t0 = wall_clock64();
    for (int i = 0; i < n; ++i) v += __hip_atomic_fetch_add(p + (v & 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t1 = wall_clock64(); out[0] = t1 - t0;
    t0 = wall_clock64();
    for (int i = 0; i < n; ++i) v += __hip_atomic_fetch_add(p + 64 + (v & 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t1 = wall_clock64(); out[1] = t1 - t0;
    t0 = wall_clock64();
    for (int i = 0; i < n; ++i) v += __hip_atomic_load(p + 128 + (v & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t1 = wall_clock64(); out[2] = t1 - t0;
    t0 = wall_clock64();
    for (int i = 0; i < n; ++i) v += *(volatile int*)(p + 192 + (v & 1));
    t1 = wall_clock64(); out[3] = t1 - t0;
    t0 = wall_clock64();
    for (int i = 0; i < n; ++i) { p[256 + (i & 1)] = v + i; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    t1 = wall_clock64(); out[4] = t1 - t0;
    t0 = wall_clock64();
    for (int i = 0; i < n; ++i) v += __hip_atomic_exchange(p + 320 + (v & 1), i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t1 = wall_clock64(); out[5] = t1 - t0;
    out[7] = v;
}
int main() {
    int* buf; long long* out; float* stream; const size_t SN = (size_t)256 << 20;
    CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&stream, SN * 4));
    CK(hipMemset(buf, 0, 1 << 20)); CK(hipMemset(stream, 0, SN * 4));
    const int n = 2000;
    for (int grid : {1, 1024}) {
        hipLaunchKernelGGL(k_lat, dim3(grid), dim3(256), 0, 0, buf, out, n, stream, SN);
        CK(hipDeviceSynchronize());
        long long h[8]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
        const char* nm[6] = {"agent atomicAdd (returning)", "workgroup atomicAdd (returning)", "sc1 load", "plain volatile load", "store + vmcnt(0)", "workgroup atomicExch (returning)"};
        printf("%s:\n", grid == 1 ? "idle chip" : "1023 blocks streaming");
        for (int k = 0; k < 6; ++k) printf("  %-34s %.3f us per op\n", nm[k], 0.01 * h[k] / n);
    }
    return 0;
}
