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

// ---- exp / log for the GrabCut GMM likelihoods and n-link weights (cv2.grabCut
// uses libm).  Fixed operation sequences so that quantised
// capacities and component assignments — and with them the final masks — are
// identical on the CPU reference path.  Accuracy ~1 ulp (tests).
__host__ __device__ inline double det_exp(double x) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double INV_LN2 = 1.44269504088896338700e+00;
    if (x != x) return x;
    if (x < -708.0) return 0.0;
    if (x > 709.0) return bits_to_double(0x7FF0000000000000ULL);
    const double k = rint(x * INV_LN2);
    const double r = (x - k * LN2_HI) - k * LN2_LO;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const long long ki = (long long)k;
    const long long k1 = ki / 2, k2 = ki - k1;
    return p * bits_to_double((uint64_t)(k1 + 1023) << 52) * bits_to_double((uint64_t)(k2 + 1023) << 52);
}

__host__ __device__ inline double det_log(double x) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    if (x != x || x < 0.0) return bits_to_double(0x7FF8000000000000ULL);
    if (x < 2.2250738585072014e-308) return bits_to_double(0xFFF0000000000000ULL);   // zero and subnormals
    if (x > 1.7976931348623157e308) return x;
    const uint64_t u = double_to_bits(x);
    long long e = (long long)((u >> 52) & 0x7FF) - 1023;
    double m = bits_to_double((u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double q = 1.0 / 23.0;
    q = q * z + 1.0 / 21.0;
    q = q * z + 1.0 / 19.0;
    q = q * z + 1.0 / 17.0;
    q = q * z + 1.0 / 15.0;
    q = q * z + 1.0 / 13.0;
    q = q * z + 1.0 / 11.0;
    q = q * z + 1.0 / 9.0;
    q = q * z + 1.0 / 7.0;
    q = q * z + 1.0 / 5.0;
    q = q * z + 1.0 / 3.0;
    q = q * z + 1.0;
    const double de = (double)e;
    return de * LN2_HI + (de * LN2_LO + 2.0 * s * q);
}

} // namespace ggc
