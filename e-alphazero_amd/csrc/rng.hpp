// rng.hpp -- counter-based random numbers of the engine (host + device).
//
// The reference draws all noise through jax.random (threefry2x32; src/selfplay.py:87,107-134,145-147), whose
// key-derivation helpers are JAX-version dependent and unpinned (Pipfile:8-15).  The engine keeps the block
// function (Threefry-2x32-20) and defines its own, documented derivation; every consumer also accepts explicit
// noise so results can be reproduced from any other generator.
//
//   fold(key, tag)    = threefry(key, ctr=(tag, 0x0EA2EA2E))              -> child key
//   bits(key, i)      = threefry(key, ctr=(i, 0)).x
//   uniform01(key, i) = bitcast((bits >> 9) | 0x3F800000) - 1            (jax.random.uniform's mantissa trick)
//   gumbel(key, i)    = -log(-log(max(tiny, u + tiny)))                   (jax.random.gumbel)
#pragma once
#include "dmath.hpp"

namespace eaz {

constexpr uint32_t kTagGumbel = 1u, kTagEnv = 2u, kTagSampleVisits = 3u, kTagSampleImproved = 4u,
                   kTagRandomAction = 5u, kTagStep = 0x100u, kTagInit = 0xFFFFu;

struct Key {
    uint32_t k0, k1;
};

EAZ_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

EAZ_HD void threefry2x32(Key k, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
    const uint32_t ks0 = k.k0, ks1 = k.k1, ks2 = k.k0 ^ k.k1 ^ 0x1BD11BDAu;
    uint32_t x0 = c0 + ks0, x1 = c1 + ks1;
#define EAZ_TF_R(r) { x0 += x1; x1 = rotl32(x1, r); x1 ^= x0; }
    EAZ_TF_R(13) EAZ_TF_R(15) EAZ_TF_R(26) EAZ_TF_R(6)
    x0 += ks1; x1 += ks2 + 1u;
    EAZ_TF_R(17) EAZ_TF_R(29) EAZ_TF_R(16) EAZ_TF_R(24)
    x0 += ks2; x1 += ks0 + 2u;
    EAZ_TF_R(13) EAZ_TF_R(15) EAZ_TF_R(26) EAZ_TF_R(6)
    x0 += ks0; x1 += ks1 + 3u;
    EAZ_TF_R(17) EAZ_TF_R(29) EAZ_TF_R(16) EAZ_TF_R(24)
    x0 += ks1; x1 += ks2 + 4u;
    EAZ_TF_R(13) EAZ_TF_R(15) EAZ_TF_R(26) EAZ_TF_R(6)
    x0 += ks2; x1 += ks0 + 5u;
#undef EAZ_TF_R
    o0 = x0;
    o1 = x1;
}

EAZ_HD Key fold(Key k, uint32_t tag) {
    Key r;
    threefry2x32(k, tag, 0x0EA2EA2Eu, r.k0, r.k1);
    return r;
}

EAZ_HD uint32_t rbits(Key k, uint32_t i) {
    uint32_t a, b;
    threefry2x32(k, i, 0u, a, b);
    return a;
}

EAZ_HD float uniform01(Key k, uint32_t i) { return u2f((rbits(k, i) >> 9) | 0x3F800000u) - 1.0f; }

EAZ_HD float gumbel(Key k, uint32_t i) {
    float u = uniform01(k, i) + kF32Tiny;
    u = u > kF32Tiny ? u : kF32Tiny;
    return -det_log(-det_log(u));
}

inline Key key_from_u64(uint64_t k) { return Key{static_cast<uint32_t>(k), static_cast<uint32_t>(k >> 32)}; }

}  // namespace eaz
