/*
 * oracle/o_math.h -- TEST INFRASTRUCTURE (parity oracle). Not shipped, not on the product path.
 *
 * Deterministic f32 math used by the CPU oracle.  The reference (JAX/XLA) uses whatever exp/log/tanh
 * XLA emits; those are not bit-specified, so the build fixes ONE specification of each transcendental
 * (DESIGN.md "Deterministic f32 math") as an explicit sequence of IEEE-754 binary32 operations
 * (fmaf, *, +, /, floorf -- every one correctly rounded on x86-64 and on gfx950).  The HIP kernels
 * implement the same specification independently (e-alphazero_amd/csrc/dmath.hpp), which is what makes
 * integer tree indices / visit counts bit-comparable between oracle and device.
 *
 * Accuracy of each function against libm is pinned in tests/test_oracle_math.py (<= 2 ulp on the tested
 * ranges), i.e. far inside the 1e-5 tolerance the north-star allows against the JAX path.
 *
 * Compile with -ffp-contract=off (no implicit contraction; every fused op below is an explicit fmaf).
 */
#ifndef EAZ_ORACLE_MATH_H
#define EAZ_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t o_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float o_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* 2^k for k in [-126, 127] */
static inline float o_pow2i(int k) { return o_u2f((uint32_t)(k + 127) << 23); }

/* exp(x): range reduction x = n*ln2 + r, |r| <= ln2/2, degree-6 polynomial (Cephes expf constants). */
static inline float o_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.972077083991796f) return 0.0f;
    float n = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    p = fmaf(p, z, r);
    p = p + 1.0f;
    int ni = (int)n;
    if (ni > 127) { p = p * o_pow2i(127); ni -= 127; }
    if (ni < -126) { p = p * o_pow2i(-100); ni += 100; }
    return p * o_pow2i(ni);
}

/* log(x) for x > 0 (Cephes logf constants). x <= 0 -> -inf (0) / NaN (<0). */
static inline float o_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    uint32_t u = o_f2u(x);
    int e = 0;
    if ((u >> 23) == 0) { /* subnormal: scale by 2^23 */
        x = x * 8388608.0f;
        u = o_f2u(x);
        e = -23;
    }
    e += (int)(u >> 23) - 126;                         /* x = m * 2^e, m in [0.5, 1) */
    float m = o_u2f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292E-2f;
    p = fmaf(p, m, -1.1514610310E-1f);
    p = fmaf(p, m, 1.1676998740E-1f);
    p = fmaf(p, m, -1.2420140846E-1f);
    p = fmaf(p, m, 1.4249322787E-1f);
    p = fmaf(p, m, -1.6668057665E-1f);
    p = fmaf(p, m, 2.0000714765E-1f);
    p = fmaf(p, m, -2.4999993993E-1f);
    p = fmaf(p, m, 3.3333331174E-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* tanh(x) (Cephes tanhf structure: odd polynomial below 0.625, 1 - 2/(exp(2|x|)+1) above). */
static inline float o_tanhf(float x) {
    if (x != x) return x;
    float a = fabsf(x);
    if (a > 44.0f) return x > 0.0f ? 1.0f : -1.0f;
    if (a >= 0.625f) {
        float s = o_expf(a + a);
        float t = 1.0f - 2.0f / (s + 1.0f);
        return x < 0.0f ? -t : t;
    }
    float z = x * x;
    float p = -5.70498872745E-3f;
    p = fmaf(p, z, 2.06390887954E-2f);
    p = fmaf(p, z, -5.37397155531E-2f);
    p = fmaf(p, z, 1.33314422036E-1f);
    p = fmaf(p, z, -3.33332819422E-1f);
    return fmaf(p * z, x, x);
}

/* Fixed-shape pairwise ("butterfly") sum over P = pow2ceil(n) slots, missing slots = +0.0f.
 * Level s adds element i and i^s for s = 1, 2, 4, ...  This is the reduction order the device
 * reproduces with in-lane adds + __shfl_xor, so every float reduction in the search is bit-comparable. */
static inline float o_tree_sum(const float* v, int n) {
    float buf[256];
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = 0; i < P; ++i) buf[i] = i < n ? v[i] : 0.0f;
    for (int s = 1; s < P; s <<= 1)
        for (int i = 0; i < P; i += 2 * s) buf[i] = buf[i] + buf[i + s];
    return buf[0];
}

#endif
