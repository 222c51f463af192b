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
