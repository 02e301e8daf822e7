// dmath.hpp -- deterministic binary32 transcendental functions for the gfx950 kernels.
//
// XLA's exp/log/tanh are not bit-specified, so the engine fixes its own specification (DESIGN.md
// "Deterministic f32 math"): range reduction + polynomial, written with explicit __builtin_fmaf / * / + / /
// only.  All of those are correctly rounded on gfx950 (HIP's default -fhip-fp32-correctly-rounded-divide-sqrt,
// denormals preserved), and the library is compiled with -ffp-contract=off, so results do not depend on the
// compiler's fusion choices.  Constants are the classic Cephes single-precision sets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eaz {

#define EAZ_HD __host__ __device__ __forceinline__

EAZ_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
EAZ_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
EAZ_HD float pow2i(int k) { return u2f(static_cast<uint32_t>(k + 127) << 23); }

constexpr float kF32Min = -3.40282347e+38f;  // jnp.finfo(float32).min (context.py:137)
constexpr float kF32Tiny = 1.17549435e-38f;  // jnp.finfo(float32).tiny
constexpr float kInf = __builtin_huge_valf();

EAZ_HD float det_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return kInf;
    if (x < -103.972077083991796f) return 0.0f;
    const float n = __builtin_floorf(__builtin_fmaf(x, 1.44269504088896341f, 0.5f));
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    p = __builtin_fmaf(p, z, r);
    p = p + 1.0f;
    int ni = static_cast<int>(n);
    if (ni > 127) { p = p * pow2i(127); ni -= 127; }
    if (ni < -126) { p = p * pow2i(-100); ni += 100; }
    return p * pow2i(ni);
}

EAZ_HD float det_log(float x) {
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -kInf;
    if (x == kInf) return x;
    uint32_t u = f2u(x);
    int e = 0;
    if ((u >> 23) == 0) {
        x = x * 8388608.0f;
        u = f2u(x);
        e = -23;
    }
    e += static_cast<int>(u >> 23) - 126;
    float m = u2f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    const float z = m * m;
    float p = 7.0376836292E-2f;
    p = __builtin_fmaf(p, m, -1.1514610310E-1f);
    p = __builtin_fmaf(p, m, 1.1676998740E-1f);
    p = __builtin_fmaf(p, m, -1.2420140846E-1f);
    p = __builtin_fmaf(p, m, 1.4249322787E-1f);
    p = __builtin_fmaf(p, m, -1.6668057665E-1f);
    p = __builtin_fmaf(p, m, 2.0000714765E-1f);
    p = __builtin_fmaf(p, m, -2.4999993993E-1f);
    p = __builtin_fmaf(p, m, 3.3333331174E-1f);
    float y = (p * m) * z;
    const float fe = static_cast<float>(e);
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

EAZ_HD float det_tanh(float x) {
    if (x != x) return x;
    const float a = __builtin_fabsf(x);
    if (a > 44.0f) return x > 0.0f ? 1.0f : -1.0f;
    if (a >= 0.625f) {
        const float s = det_exp(a + a);
        const float t = 1.0f - 2.0f / (s + 1.0f);
        return x < 0.0f ? -t : t;
    }
    const float z = x * x;
    float p = -5.70498872745E-3f;
    p = __builtin_fmaf(p, z, 2.06390887954E-2f);
    p = __builtin_fmaf(p, z, -5.37397155531E-2f);
    p = __builtin_fmaf(p, z, 1.33314422036E-1f);
    p = __builtin_fmaf(p, z, -3.33332819422E-1f);
    return __builtin_fmaf(p * z, x, x);
}

// fully_connected.py:92-96 (not is_training): u = clip(max(scaled_state_novelty, u * max_u), 0, max_u); `u_scaled` = u * max_u
EAZ_HD float ube_combine(float u_scaled, float novelty, float max_u) {
    float u = novelty > u_scaled ? novelty : u_scaled;
    u = u < 0.0f ? 0.0f : u;
    return u > max_u ? max_u : u;
}

}  // namespace eaz
