// Micro-test: in which order, and with which roundings, does v_mfma_f32_32x32x2_f32 accumulate its two k values?
// D[i][j] = A[i][k0] B[k0][j] + A[i][k1] B[k1][j] + C[i][j], k0 from lanes 0..31, k1 from lanes 32..63.
// Candidates (all in float32): fma(a1,b1, fma(a0,b0,c)), fma(a0,b0, fma(a1,b1,c)), (a0 b0 + a1 b1 rounded) + c, exact sum rounded once.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* a0, const float* a1, const float* b0, const float* b1, const float* c, float* out) {
    const int lane = threadIdx.x, hk = lane >> 5, li = lane & 31;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c[((r & 3) + 8 * (r >> 2) + 4 * hk) * 32 + li];
    const float a = hk ? a1[li] : a0[li];      // A operand: row li, k = hk
    const float b = hk ? b1[li] : b0[li];      // B operand: column li, k = hk
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * hk) * 32 + li] = acc[r];
}

int main() {
    srand(1);
    std::vector<float> a0(32), a1(32), b0(32), b1(32), c(1024), o(1024);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX - 0.5) * 8.0); };
    for (int i = 0; i < 32; ++i) { a0[i] = rnd(); a1[i] = rnd(); b0[i] = rnd(); b1[i] = rnd(); }
    for (auto& v : c) v = rnd() * 1e-3f;
    float *d[6];
    for (int i = 0; i < 6; ++i) hipMalloc(&d[i], 4096);
    hipMemcpy(d[0], a0.data(), 128, hipMemcpyHostToDevice); hipMemcpy(d[1], a1.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(d[2], b0.data(), 128, hipMemcpyHostToDevice); hipMemcpy(d[3], b1.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(d[4], c.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5]);
    hipMemcpy(o.data(), d[5], 4096, hipMemcpyDeviceToHost);
    int m[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            const float cc = c[i * 32 + j], got = o[i * 32 + j];
            const float v0 = fmaf(a1[i], b1[j], fmaf(a0[i], b0[j], cc));
            const float v1 = fmaf(a0[i], b0[j], fmaf(a1[i], b1[j], cc));
            const float v2 = (float)((double)a0[i] * b0[j] + (double)a1[i] * b1[j] + (double)cc);   // ~exact, one rounding
            const float p0 = a0[i] * b0[j], p1 = a1[i] * b1[j];
            const float v3 = (cc + p0) + p1;
            m[0] += got == v0; m[1] += got == v1; m[2] += got == v2; m[3] += got == v3;
            m[4] += (v0 != v1);
        }
    printf("of 1024 outputs: fma(k1, fma(k0, c)) matches %d; fma(k0, fma(k1, c)) matches %d; exact-sum-rounded-once matches %d; "
           "separate mul+add matches %d; (the two fma orders differ on %d)\n", m[0], m[1], m[2], m[3], m[4]);
    return 0;
}
