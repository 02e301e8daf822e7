/*
 * oracle/o_rng.h -- TEST INFRASTRUCTURE (parity oracle).
 *
 * Counter-based RNG of the build.  The reference draws every random number through jax.random
 * (threefry2x32; src/selfplay.py:87,107-134,145-147, src/context.py:126) whose split/uniform derivations are
 * version dependent and unpinned (Pipfile:8-15), so the build does NOT claim bit parity with JAX key
 * streams.  Instead: the primitive is the same block function (Threefry-2x32, 20 rounds,
 * pinned by the Random123 known-answer vectors in tests/test_oracle_rng.py), the derivation scheme below is
 * the build's own and is identical in oracle and device, and every consumer of noise (root Gumbel, reset
 * keys, action sampling) also accepts explicit arrays so parity is conditional on identical noise.
 *
 *   fold(key, tag)        = threefry(key, ctr = (tag, 0x0EA2EA2E))            -> new 64-bit key
 *   bits(key, i)          = threefry(key, ctr = (i, 0))[0]
 *   uniform01(key, i)     = bitcast((bits >> 9) | 0x3F800000) - 1.0f         in [0,1)   (jax.random.uniform's mantissa trick)
 *   gumbel(key, i)        = -log(-log(max(tiny, u + tiny)))                   (jax.random.gumbel: uniform(minval=tiny, maxval=1))
 */
#ifndef EAZ_ORACLE_RNG_H
#define EAZ_ORACLE_RNG_H

#include "o_math.h"

#define O_TAG_GUMBEL 1u
#define O_TAG_ENV 2u
#define O_TAG_SAMPLE_VISITS 3u
#define O_TAG_SAMPLE_IMPROVED 4u
#define O_TAG_RANDOM_ACTION 5u
#define O_TAG_STEP 0x100u /* + t */

static inline uint32_t o_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static inline void o_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2]) {
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
    for (int blk = 0; blk < 5; ++blk) {
        const int* rr = R[blk & 1];
        for (int i = 0; i < 4; ++i) {
            x0 += x1;
            x1 = o_rotl32(x1, rr[i]);
            x1 ^= x0;
        }
        x0 += ks[(blk + 1) % 3];
        x1 += ks[(blk + 2) % 3] + (uint32_t)(blk + 1);
    }
    out[0] = x0;
    out[1] = x1;
}

typedef struct { uint32_t k0, k1; } OKey;

static inline OKey o_fold(OKey k, uint32_t tag) {
    uint32_t o[2];
    o_threefry2x32(k.k0, k.k1, tag, 0x0EA2EA2Eu, o);
    OKey r = {o[0], o[1]};
    return r;
}

static inline uint32_t o_bits(OKey k, uint32_t i) {
    uint32_t o[2];
    o_threefry2x32(k.k0, k.k1, i, 0u, o);
    return o[0];
}

static inline float o_uniform01(OKey k, uint32_t i) {
    return o_u2f((o_bits(k, i) >> 9) | 0x3F800000u) - 1.0f;
}

static inline float o_gumbel(OKey k, uint32_t i) {
    const float tiny = 1.17549435e-38f;
    float u = o_uniform01(k, i) + tiny;
    u = u > tiny ? u : tiny;
    return -o_logf(-o_logf(u));
}

#endif
