/*
 * oracle/mathfn.c — deterministic cube root and x^2.4 built from +,-,*,/ only.
 * TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * skimage's rgb2lab (colorconv.py:657-662, 952-955 of scikit-image 0.18.3)
 * calls np.power(., 2.4) and np.cbrt, whose last-ulp behaviour depends on the
 * libm / SIMD build.  Integer decisions downstream (SLIC labels) must agree
 * bit-for-bit between this oracle and the HIP kernels, so both restate the two
 * functions as the same fixed sequence of IEEE-754 double operations:
 * an exponent-arithmetic seed followed by a fixed number of Newton steps.
 * Agreement with libm is checked to 1e-14 relative in tests/test_color_oracle.py.
 */
#include "ggc_oracle.h"
#include <string.h>

static double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static uint64_t to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

/* cube root of a > 0 (normal range) */
double ggo_cbrt(double a) {
    /* seed: divide the biased exponent by 3 (relative error < 6%) */
    double y = from_bits(to_bits(a) / 3 + 0x2A9F7893782DA1CEULL);
    for (int i = 0; i < 6; ++i) {
        double y2 = y * y;
        y = y - (y2 * y - a) / (3.0 * y2);
    }
    return y;
}

/* fifth root of q > 0 */
static double root5(double q) {
    double y = from_bits(to_bits(q) / 5 + 0x3325FFFFFFFFFFFFULL);
    for (int i = 0; i < 7; ++i) {
        double y2 = y * y;
        double y4 = y2 * y2;
        y = y - (y4 * y - q) / (5.0 * y4);
    }
    return y;
}

/* a^2.4 = a^2 * (a^2)^(1/5), a > 0 */
double ggo_pow24(double a) {
    double q = a * a;
    return q * root5(q);
}

/* ---- deterministic exp / log for the GrabCut GMM likelihoods and n-link
 * weights (cv2.grabCut uses libm exp/log).  Same fixed
 * operation sequences as the HIP side so that integer capacities, component
 * assignments and therefore the final masks agree bit for bit.            */
static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
static const double INV_LN2 = 1.44269504088896338700e+00;
double rint(double);

double ggo_exp(double x) {
    if (x != x) return x;
    if (x < -708.0) return 0.0;        /* flushed: below ~1e-308 nothing downstream can tell */
    if (x > 709.0) return 1.0 / 0.0;
    const double k = rint(x * INV_LN2);
    const double r = (x - k * LN2_HI) - k * LN2_LO;
    /* exp(r), |r| <= 0.347: Taylor polynomial of degree 13, Horner */
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
    /* scale by 2^k in two exact steps (k in [-1022, 1023]) */
    const long long ki = (long long)k;
    const long long k1 = ki / 2, k2 = ki - k1;
    return p * from_bits((uint64_t)(k1 + 1023) << 52) * from_bits((uint64_t)(k2 + 1023) << 52);
}

double ggo_log(double x) {
    if (x != x || x < 0.0) return 0.0 / 0.0;
    if (x < 2.2250738585072014e-308) return -1.0 / 0.0;   /* zero and subnormals */
    if (x > 1.7976931348623157e308) return x;
    uint64_t u = to_bits(x);
    long long e = (long long)((u >> 52) & 0x7FF) - 1023;
    double m = from_bits((u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL);  /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }                         /* [sqrt(1/2), sqrt(2)) */
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
