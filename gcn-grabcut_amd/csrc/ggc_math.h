// ggc_math.h — deterministic cube root and x^2.4 for the colour conversions.
//
// skimage's rgb2lab (the reference calls it at graph_builder.py:148 and again,
// in float32, inside slic) uses np.power(., 2.4) and np.cbrt, whose last ulp
// depends on the libm build.  The SLIC label map must be bit-identical between
// this library and the CPU reference path, so both sides define the two
// functions as the same fixed sequence of IEEE-754 double operations
// (exponent-arithmetic seed + a fixed number of Newton steps; +,-,*,/ only,
// compiled with -ffp-contract=off).  Accuracy: <= 1 ulp of f64 (tests).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

namespace ggc {

__host__ __device__ inline double bits_to_double(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d; std::memcpy(&d, &u, 8); return d;
#endif
}
__host__ __device__ inline uint64_t double_to_bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u; std::memcpy(&u, &d, 8); return u;
#endif
}

// cube root of a > 0 (normal range)
__host__ __device__ inline double det_cbrt(double a) {
    double y = bits_to_double(double_to_bits(a) / 3 + 0x2A9F7893782DA1CEULL);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double y2 = y * y;
        y = y - (y2 * y - a) / (3.0 * y2);
    }
    return y;
}

// a^2.4 = a^2 * (a^2)^(1/5), a > 0
__host__ __device__ inline double det_pow24(double a) {
    const double q = a * a;
    double y = bits_to_double(double_to_bits(q) / 5 + 0x3325FFFFFFFFFFFFULL);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double y2 = y * y;
        const double y4 = y2 * y2;
        y = y - (y4 * y - q) / (5.0 * y4);
    }
    return q * y;
}

} // namespace ggc
