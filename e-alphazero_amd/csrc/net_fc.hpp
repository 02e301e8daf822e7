// net_fc.hpp -- EpistemicFullyConnectedAZNet forward (src/network/fully_connected.py:41-101) and the XXHash-variant
// novelty set (src/network/hashes.py:23-50,162-229) for gfx950.
//
// Numerics contract (DESIGN.md "FC forward"): hk.Linear is y = x @ w + b with, per output element,
//     acc = +0;  acc = fmaf(x[k], w[k][j], acc) for k = 0..K-1;  y = acc + b[j]
// which is bit-for-bit what a chain of v_mfma_f32_16x16x4_f32 instructions computes (f32 in, f32 accumulate, one
// rounding per product, k ascending).  Layers 1 and 2 of every head run on that MFMA; layer 3 (out = 1 or A) is a
// VALU fmaf chain in the same order.  tanh is det_tanh (dmath.hpp).
//
// Kernel shape: one 256-thread workgroup = 64 batch rows x one head.  The three layers of the head stay on chip:
//   layer 1: A = observation tile, staged ONCE per 1024 inputs as uint8 into LDS (K-major, 4x4 byte transposes in
//            registers); B = W1 rows streamed from L2 into registers.  Only k-steps (4 consecutive inputs) in which the
//            64-row tile has a non-zero entry are executed -- skipping an all-zero k-step is exact because
//            fmaf(0, w, acc) == acc -- so one-hot / bit-encoded observations cost a fraction of the dense GEMM.
//   layer 2: A = relu(h1) kept in LDS (K-major f32), B = W2 from L2, dense.
//   layer 3: VALU dot products over relu(h2) in LDS, then the head's activation.
// Each wave owns 64 output columns x 64 rows = 4x4 MFMA tiles (64 accumulator registers).  Rows and columns are
// interleaved across tiles (row = 4*i + s, col = 4*j + t) so that both operands are fetched with one 4/16-byte
// access per lane per k-step.  Weights are prefetched one group of 8 k-steps (32 VGPRs) ahead of the MFMAs.
// blockIdx -> (head, row tile) keeps all workgroups of an XCD on as few heads as possible, so a head's 1.25 MB of
// weights stays resident in that XCD's 4 MiB L2.
#pragma once
#include "dmath.hpp"
#include "rng.hpp"

namespace eaz {

struct NetDev {
    int in_dim, H, A;
    const float* w[12];  // device, haiku order: value 0-2, ube 3-5, exploitation 6-8, exploration 9-11; w:[in,out]
    const float* b[12];
    float max_u, nov_scale;
    int hash_off, hash_len, hash_bits;
    uint8_t* bitset;  // device, 2^(hash_bits-3) bytes
};

struct NetOut {
    float* exploit;        // [B,A] or null
    float* explore;        // [B,A] or null
    float* value;          // [B] or null
    float* ube;            // [B] or null: 0.5 * (tanh + 1) * max_u, BEFORE the novelty max / clip (see ube_combine)
};

__global__ void __launch_bounds__(256) ube_combine_kernel(float* ube, const float* novelty, float max_u, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) ube[b] = ube_combine(ube[b], novelty[b], max_u);
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kFcM = 64;      // batch rows per workgroup
constexpr int kFcKC = 1024;   // layer-1 inputs staged per barrier (uint8 tile [KC][68]); = 4 mask blocks of 256
constexpr int kFcLD = 68;     // leading dimension of the K-major LDS tiles: 64 rows + 4 pad (floats for H1/H2, bytes for the obs tile)
constexpr int kFcH = 256;     // hidden width the kernel is specialised for (config.py:24)
constexpr int kFcG = 8;       // k-steps per weight-prefetch group

// LDS map (bytes): [0, 69632) H1 = relu(layer 1) f32 [256][68];  [69632, 139264) region 2 = the uint8 observation chunk
// [1024][68] during layer 1, then H2 = relu(layer 2) f32 [256][68];  then one 64-bit "non-zero k-step" mask per 256 inputs.
constexpr int kFcH1Bytes = kFcH * kFcLD * 4;
constexpr int kFcMaxBlocks = 64;  // in_dim <= 16384
constexpr int kFcW3Cols = 16;     // layer-3 weights of heads with <= 16 outputs are staged in LDS ([256][No] floats)
constexpr int kFcLdsBytes = 2 * kFcH1Bytes + 8 * kFcMaxBlocks + kFcH * kFcW3Cols * 4;
static_assert(kFcKC * kFcLD == kFcH1Bytes, "the observation chunk aliases the H2 tile exactly");

// head_list: up to 4 head ids (0 value, 1 ube, 2 exploitation, 3 exploration) to evaluate
struct HeadList {
    int n;
    int id[4];
};

// Observation chunk in flight: thread (kq4 = tid & 63, rg = tid >> 6) holds, for each of the 4 blocks of 256 inputs,
// inputs 4*kq4..+3 of rows 16*rg..+15 as 16 packed words.
struct AStage {
    uint32_t w[4][4][4];  // [block][j][i]: row 16*rg + 4*j + i
};

__device__ __forceinline__ void a_stage_load(AStage& s, const uint8_t* obs, int B, int in_dim, int row0, int chunk, int tid, bool fast) {
    const int kq4 = tid & 63, rg = tid >> 6;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int k = chunk * kFcKC + kb * 256 + 4 * kq4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 16 * rg + 4 * j + i;
                uint32_t v = 0;
                if (row < B && k < in_dim) {
                    const uint8_t* p = obs + static_cast<long>(row) * in_dim + k;
                    if (fast) v = *reinterpret_cast<const uint32_t*>(p);
                    else
                        for (int q = 0; q < 4; ++q) v |= (k + q < in_dim ? static_cast<uint32_t>(p[q]) : 0u) << (8 * q);
                }
                s.w[kb][j][i] = v;
            }
    }
}

// registers -> LDS uint8 [k][row] (4x4 byte transposes) + per-block masks of the k-steps that hold a non-zero entry
__device__ __forceinline__ void a_stage_store(uint8_t* At, unsigned long long* kmask, const AStage& s, int tid) {
    const int kq4 = tid & 63, rg = tid >> 6;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w0 = s.w[kb][j][0], w1 = s.w[kb][j][1], w2 = s.w[kb][j][2], w3 = s.w[kb][j][3];
            any |= w0 | w1 | w2 | w3;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint32_t o = ((w0 >> (8 * kk)) & 0xFFu) | (((w1 >> (8 * kk)) & 0xFFu) << 8) |
                                   (((w2 >> (8 * kk)) & 0xFFu) << 16) | (((w3 >> (8 * kk)) & 0xFFu) << 24);
                *reinterpret_cast<uint32_t*>(At + (kb * 256 + 4 * kq4 + kk) * kFcLD + 16 * rg + 4 * j) = o;
            }
        }
        const unsigned long long m = __ballot(any != 0);  // lane = k-step index within the block
        if ((tid & 63) == 0 && m) atomicOr(&kmask[kb], m);
    }
}

#define EAZ_MFMA16(acc, a, b)                                                              \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) \
        acc[s_][t_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], b[t_], acc[s_][t_], 0, 0, 0)

// accumulators -> relu(acc + bias) -> LDS K-major tile Ht[col][row]
__device__ __forceinline__ void store_hidden(float* Ht, const floatx4 (&acc)[4][4], const float* bias, int n0, int r, int kq) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = n0 + 4 * r + t;
        const float bj = bias[col];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            floatx4 o;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float v = acc[s][t][reg] + bj;
                o[s] = v > 0.0f ? v : 0.0f;
            }
            // rows 16*kq + 4*reg + s, s = 0..3
            *reinterpret_cast<floatx4*>(Ht + col * kFcLD + 16 * kq + 4 * reg) = o;
        }
    }
}

__device__ __forceinline__ floatx4 cvt_rows(uint32_t packed) {
    return floatx4{static_cast<float>(packed & 0xFFu), static_cast<float>((packed >> 8) & 0xFFu),
                   static_cast<float>((packed >> 16) & 0xFFu), static_cast<float>(packed >> 24)};
}

// up to kFcG executed k-steps (wave-uniform indices inside a 256-input block) and their weight fragments
struct KGroup {
    int n;
    int ks[kFcG];
};
struct BGroup {
    floatx4 v[kFcG];
};

__device__ __forceinline__ KGroup next_group(unsigned long long& mask) {
    KGroup g;
    g.n = 0;
#pragma unroll
    for (int i = 0; i < kFcG; ++i) {
        g.ks[i] = 0;
        if (mask) {
            g.ks[i] = __builtin_ctzll(mask);
            mask &= mask - 1;
            g.n = i + 1;
        }
    }
    return g;
}

__device__ __forceinline__ void load_group(BGroup& b, const KGroup& g, const float* W, int K, int k0, int col, int kq) {
#pragma unroll
    for (int i = 0; i < kFcG; ++i) {
        const int k = k0 + 4 * g.ks[i] + kq;
        if (i < g.n && k < K) b.v[i] = *reinterpret_cast<const floatx4*>(W + static_cast<long>(k) * kFcH + col);
        else b.v[i] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

// One block of 256 inputs (64 k-steps): executes the k-steps set in `mask`, weights one group ahead.
// U8A: A tile is the uint8 observation chunk (layer 1), else the f32 hidden tile (layer 2).
template <bool U8A>
__device__ __forceinline__ void mma_block(floatx4 (&acc)[4][4], const void* Abase, unsigned long long mask, const float* W, int K, int k0,
                                          int col, int r, int kq) {
    KGroup gc = next_group(mask);
    if (gc.n == 0) return;
    BGroup bc, bn;
    load_group(bc, gc, W, K, k0, col, kq);
    while (true) {
        const KGroup gn = next_group(mask);
        if (gn.n) load_group(bn, gn, W, K, k0, col, kq);
#pragma unroll
        for (int i = 0; i < kFcG; ++i) {
            if (i < gc.n) {
                floatx4 a;
                if constexpr (U8A) {
                    const uint8_t* At = static_cast<const uint8_t*>(Abase);
                    a = cvt_rows(*reinterpret_cast<const uint32_t*>(At + (4 * gc.ks[i] + kq) * kFcLD + 4 * r));
                } else {
                    const float* At = static_cast<const float*>(Abase);
                    a = *reinterpret_cast<const floatx4*>(At + (4 * gc.ks[i] + kq) * kFcLD + 4 * r);
                }
                EAZ_MFMA16(acc, a, bc.v[i]);
            }
        }
        if (gn.n == 0) break;
        gc = gn;
        bc = bn;
    }
}

__global__ void __launch_bounds__(256, 1) fc_forward_kernel(NetDev net, const uint8_t* __restrict__ obs, int B, NetOut out, HeadList heads,
                                                            unsigned long long* kstep_counter) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* H1 = reinterpret_cast<float*>(lds);
    uint8_t* R2 = lds + kFcH1Bytes;
    float* H2 = reinterpret_cast<float*>(R2);
    unsigned long long* kmask = reinterpret_cast<unsigned long long*>(lds + 2 * kFcH1Bytes);

    // blockIdx -> (head, row tile): XCD x (= blockIdx % 8 under round-robin dispatch) serves head x % n only
    const int nh = heads.n;
    const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
    const int hsel = xcd % nh;
    int c_h = 0;
    for (int x = 0; x < 8; ++x) c_h += (x % nh) == hsel;
    const int tile = grp * c_h + xcd / nh;
    const int n_tiles = (B + kFcM - 1) / kFcM;
    if (tile >= n_tiles) return;
    const int head = heads.id[hsel];
    const int row0 = tile * kFcM;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int n0 = wave * 64, col = n0 + 4 * r;
    const int K1 = net.in_dim;
    const float *W1 = net.w[head * 3], *W2 = net.w[head * 3 + 1], *W3 = net.w[head * 3 + 2];
    const float *b1 = net.b[head * 3], *b2 = net.b[head * 3 + 1], *b3 = net.b[head * 3 + 2];
    const bool fast = (K1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 3) == 0);

    floatx4 acc[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---------------- layer 1: [64 x K1] x [K1 x 256], all-zero k-steps skipped ----------------
    const int nchunk1 = (K1 + kFcKC - 1) / kFcKC;
    unsigned long long executed = 0;
    if (tid < kFcMaxBlocks) kmask[tid] = 0ull;
    // layer-3 weights: [256][No] floats, staged now (coalesced) and read back as LDS broadcasts after layer 2
    const int No = head < 2 ? 1 : net.A;
    float* W3s = reinterpret_cast<float*>(lds + 2 * kFcH1Bytes + 8 * kFcMaxBlocks);
    const bool w3_lds = No <= kFcW3Cols;
    if (w3_lds)
        for (int i = tid; i < kFcH * No; i += 256) W3s[i] = W3[i];
    AStage st;
    a_stage_load(st, obs, B, K1, row0, 0, tid, fast);
    __syncthreads();  // masks zeroed
    for (int c = 0; c < nchunk1; ++c) {
        a_stage_store(R2, &kmask[4 * c], st, tid);
        __syncthreads();  // chunk c and its masks are visible
        if (c + 1 < nchunk1) a_stage_load(st, obs, B, K1, row0, c + 1, tid, fast);  // in flight during the MFMAs below
#pragma unroll 1
        for (int kb = 0; kb < 4; ++kb) {
            const unsigned long long mv = kmask[4 * c + kb];
            // the mask is the same in every lane: move it to scalar registers so the k-step loop is scalar control flow
            const unsigned long long m = static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(mv))) |
                                         (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(mv >> 32))) << 32);
            executed += __builtin_popcountll(m);
            mma_block<true>(acc, R2 + kb * 256 * kFcLD, m, W1, K1, c * kFcKC + kb * 256, col, r, kq);
        }
        if (c + 1 < nchunk1) __syncthreads();  // everyone is done reading chunk c before it is overwritten
    }
    store_hidden(H1, acc, b1, n0, r, kq);
    __syncthreads();  // H1 complete; region 2 (observation chunk) is dead from here on

    // ---------------- layer 2: [64 x 256] x [256 x 256], dense ----------------
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    mma_block<false>(acc, H1, ~0ull, W2, kFcH, 0, col, r, kq);
    store_hidden(H2, acc, b2, n0, r, kq);
    __syncthreads();

    // ---------------- layer 3 + head activation (VALU, k ascending) ----------------
    const int row = tid & 63;
    const bool row_ok = row0 + row < B;
    for (int o = wave; o < No; o += 4) {
        float z = 0.0f;
        if (w3_lds) {
#pragma unroll 16
            for (int k = 0; k < kFcH; ++k) z = __builtin_fmaf(H2[k * kFcLD + row], W3s[k * No + o], z);
        } else {
#pragma unroll 16
            for (int k = 0; k < kFcH; ++k) z = __builtin_fmaf(H2[k * kFcLD + row], W3[static_cast<long>(k) * No + o], z);
        }
        z = z + b3[o];
        if (!row_ok) continue;
        const long gi = row0 + row;
        if (head == 0) {
            if (out.value) out.value[gi] = det_tanh(z);                       // fully_connected.py:55
        } else if (head == 1) {
            float u = 0.5f * (det_tanh(z) + 1.0f);                            // :64
            u = u * net.max_u;                                                // :93
            // :95-96 u = clip(max(novelty, u), 0, max_u) is applied by ube_combine() where the novelty lookup (an
            // independent kernel that runs concurrently with this one) is joined
            if (out.ube) out.ube[gi] = u;
        } else {
            float* dst = head == 2 ? out.exploit : out.explore;               // :68-81
            if (dst) dst[gi * net.A + o] = z;
        }
    }
    if (kstep_counter && tid == 0) atomicAdd(kstep_counter, executed);
}

// ---- XXHash-variant novelty (src/network/hashes.py:162-229): 4 consecutive lanes per sample, lane i = quarter i ----
// mode: 0 = lookup (writes novelty/seen/indices), 1 = update (scatter-OR, hashes.py:45-50)
__device__ __forceinline__ uint32_t xxh_round_word(uint32_t acc, uint32_t w) {
    const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t lane_val = f2u(static_cast<float>((w >> (8 * j)) & 0xFFu));  // astype(f32) + bitcast, hashes.py:207
        acc = rotl32(acc + lane_val * P2, 13) * P1;                                 // :176-181
    }
    return acc;
}

__global__ void __launch_bounds__(256) hash_kernel(NetDev net, const uint8_t* __restrict__ obs, int B, float* novelty,
                                                   uint8_t* seen_out, uint32_t* idx_out, int mode) {
    const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, SEED = 1u;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gtid >> 2, i = gtid & 3;
    const bool live = b < B;
    const int L = net.hash_len >> 2;
    const uint8_t* p = obs + static_cast<long>(live ? b : 0) * net.in_dim + net.hash_off + i * L;
    uint32_t acc = i == 0 ? SEED + P1 + P2 : (i == 1 ? SEED + P2 : (i == 2 ? SEED : SEED - P1));  // :217-220
    if (live) {
        int l = 0;
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            // 16-byte loads, the next one in flight while the current one is consumed
            uint4 cur = make_uint4(0, 0, 0, 0);
            if (L >= 16) cur = *reinterpret_cast<const uint4*>(p);
            for (; l + 16 <= L; l += 16) {
                uint4 nxt = make_uint4(0, 0, 0, 0);
                if (l + 32 <= L) nxt = *reinterpret_cast<const uint4*>(p + l + 16);
                acc = xxh_round_word(acc, cur.x);
                acc = xxh_round_word(acc, cur.y);
                acc = xxh_round_word(acc, cur.z);
                acc = xxh_round_word(acc, cur.w);
                cur = nxt;
            }
        }
        if ((reinterpret_cast<uintptr_t>(p + l) & 3) == 0)
            for (; l + 4 <= L; l += 4) acc = xxh_round_word(acc, *reinterpret_cast<const uint32_t*>(p + l));
        for (; l < L; ++l) acc = rotl32(acc + f2u(static_cast<float>(p[l])) * P2, 13) * P1;
    }
    const int rot = i == 0 ? 1 : (i == 1 ? 7 : (i == 2 ? 12 : 18));  // :184-187
    uint32_t h = rotl32(acc, rot);
    h += __shfl_xor(h, 1);
    h += __shfl_xor(h, 2);
    h += static_cast<uint32_t>(L);  // :226 (stripe count, not byte length)
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;  // :190-197
    const uint32_t index = h >> (32 - net.hash_bits);             // :229
    if (!live || i != 0) return;
    if (mode == 1) {
        // binary_set[index >> 3] |= 1 << (index & 7)  -> 32-bit atomic OR on the containing word
        uint32_t* word = reinterpret_cast<uint32_t*>(net.bitset) + (index >> 5);
        atomicOr(word, 1u << (index & 31u));
        return;
    }
    const bool seen = net.bitset ? ((net.bitset[index >> 3] >> (index & 7u)) & 1u) != 0 : false;  // hashes.py:29-38
    if (novelty) novelty[b] = (seen ? 0.0f : 1.0f) * net.nov_scale;                               // fully_connected.py:90
    if (seen_out) seen_out[b] = seen ? 1 : 0;
    if (idx_out) idx_out[b] = index;
}

}  // namespace eaz
