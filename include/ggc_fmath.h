/* ggc_fmath.h — float32 exp / sigmoid / GELU as FIXED sequences of IEEE-754 operations (add, multiply, fused multiply-add,
 * correctly rounded divide, round-to-nearest-even, integer bit moves).  The same source is compiled into the gfx950 kernels
 * (gcn-grabcut_amd/csrc) and into the CPU oracle (oracle/), so both produce the same bits: libm's / ocml's expf and erff, and
 * the hardware's v_exp_f32 / v_rcp_f32, are each accurate to an ulp or so but not identical to one another, and a last-ulp
 * difference in a ResGCNNet probability is enough to move a trimap pixel across its threshold.
 *
 * Accuracy (against double precision): ggc_expf 2.7 ulp on [-87, 87];
 * ggc_geluf absolute error 4e-7 on [-12, 12], the same as 0.5 x (1 + erff(x / sqrt 2)) evaluated in float32
 * (reference model.py uses nn.GELU(), the exact erf form).
 *
 * Both compilers must keep one rounding per written operation: -ffp-contract=off (the Makefiles set it); fmaf is written out
 * where a fused operation is meant. */
#ifndef GGC_FMATH_H
#define GGC_FMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GGC_FM_FN __host__ __device__ __forceinline__
#else
#define GGC_FM_FN static inline
#endif

GGC_FM_FN float ggc_fm_from_bits(uint32_t u) { float f; memcpy(&f, &u, sizeof f); return f; }
GGC_FM_FN uint32_t ggc_fm_to_bits(float f) { uint32_t u; memcpy(&u, &f, sizeof u); return u; }

/* 1 / d for a positive normal d (used on [1, 7e37]): integer seed (12 % off) and three Newton steps r <- r (2 - d r), each one fused multiply-add and
 * one multiply; relative error 1.2e-7.  Not the correctly rounded quotient, but the same bits everywhere, and a third of the
 * instructions of an IEEE divide on the GPU (it vectorises on the packed-f32 pipe). */
GGC_FM_FN float ggc_rcp_nr(float d) {
    float r = ggc_fm_from_bits(0x7EF311C7u - ggc_fm_to_bits(d));
    r = r * fmaf(-d, r, 2.0f);
    r = r * fmaf(-d, r, 2.0f);
    r = r * fmaf(-d, r, 2.0f);
    return r;
}

/* exp(x), x clamped to [-87, 87] (results stay normal): n = rint(x log2 e), r = x - n ln 2 (two-piece constant), degree-6
 * Horner polynomial for exp(r) on |r| <= ln 2 / 2, then the exponent is added to the float's bits. */
GGC_FM_FN float ggc_expf(float x) {
    x = x < -87.0f ? -87.0f : (x > 87.0f ? 87.0f : x);
    /* n = rint(x log2 e) by adding and subtracting 1.5 * 2^23 (round to nearest even at integer granularity); the sum's low
     * mantissa bits hold n in two's complement, so shifting them to the exponent field gives 2^n's bits without a conversion */
    const float z = x * 1.44269504088896341f + 12582912.0f;
    const float n = z - 12582912.0f;
    float r = fmaf(n, -0.693145751953125f, x);             /* ln 2, high part (exact product for |n| < 2^11) */
    r = fmaf(n, -1.42860682030941723212e-6f, r);           /* ln 2, low part */
    float p = 1.3888889225e-3f;                            /* 1/720 ... 1/2 */
    p = fmaf(p, r, 8.3333337680e-3f);
    p = fmaf(p, r, 4.1666667908e-2f);
    p = fmaf(p, r, 1.6666667163e-1f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return p * ggc_fm_from_bits((ggc_fm_to_bits(z) << 23) + 0x3F800000u);
}

GGC_FM_FN float ggc_sigmoidf(float x) { return 1.0f / (1.0f + ggc_expf(-x)); }
/* the same with the Newton reciprocal (relative error 2e-7): the per-edge gates of GCNTrimapNet / GATTrimapNet evaluate 2e8 of
 * these per block, inside an MFMA kernel */
GGC_FM_FN float ggc_sigmoid_nr(float x) { return ggc_rcp_nr(1.0f + ggc_expf(-x)); }

/* GELU(x) = x Phi(x), Phi through erfc: with t = |x| / sqrt 2 and u = 1 / (1 + p t),
 *     erfc(t) ~= (a1 u + ... + a6 u^6) exp(-t^2),  |error| < 8e-9 on [0, 6.5]
 * (Lawson-weighted least squares against scipy.special.erfc; p = 0.39030933). */
GGC_FM_FN float ggc_geluf(float x) {
    const float u = ggc_rcp_nr(fmaf(fabsf(x), 0.27599036693573f, 1.0f));
    float q = -0.11346635967493057f;                       /* a_k / 2, highest power first */
    q = fmaf(q, u, 0.44092419743537903f);
    q = fmaf(q, u, -0.3140281140804291f);
    q = fmaf(q, u, 0.3222678005695343f);
    q = fmaf(q, u, 0.04667610302567482f);
    q = fmaf(q, u, 0.1176263764500618f);
    const float hq = (q * u) * ggc_expf((x * x) * -0.5f);  /* erfc(|x| / sqrt 2) / 2 */
    return x * (x >= 0.0f ? 1.0f - hq : hq);
}

/* Unnormalised taps exp(-0.5 i^2 / sigma^2), i = -r..r, of scipy.ndimage's Gaussian (scipy/ndimage/filters.py, _gaussian_kernel1d).
 * scipy evaluates them with numpy's vectorised exp, which differs from libm's in the last bit for three of the five values at
 * sigma = 1 — invisible after the float32 rounding of the Lab path, visible in the float64 path (use_lab=False).  For sigma = 1,
 * the setting of the reference's config, the values are the ones numpy 1.26 returns; other sigmas use libm (last-bit unpinned). */
static inline void ggc_gaussian_taps(double sigma, int r, double* w /*[2r+1]*/) {
    static const double s1[5] = {0x1.0000000000000p+0, 0x1.368b2fc6f960ap-1, 0x1.152aaa3bf81cbp-3, 0x1.6c0504695c418p-7, 0x1.5fc21041027acp-12};
    const double s2 = sigma * sigma;
    for (int i = -r; i <= r; ++i) {
        const int a = i < 0 ? -i : i;
        w[i + r] = (sigma == 1.0 && a < 5) ? s1[a] : exp(-0.5 / s2 * (double)(i * i));
    }
}

#endif /* GGC_FMATH_H */
