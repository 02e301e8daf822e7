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
//   layer 1: A = observation tile (uint8 -> f32, staged through LDS in 32-deep K chunks, stored K-major),
//            B = W1 streamed from L2 straight into registers (float4 per lane per k-step, double-buffered per chunk)
//   layer 2: A = relu(h1) kept in LDS (K-major), B = W2 from L2
//   layer 3: VALU dot products over relu(h2) in LDS, then the head's activation.
// Each wave owns 64 output columns x 64 rows = 4x4 MFMA tiles (64 accumulator VGPRs).  Rows and columns are
// interleaved across tiles (row = 4*i + s, col = 4*j + t) so that both operands are fetched with one 16-byte
// access per lane per k-step.  blockIdx -> (head, row tile) keeps all workgroups of an XCD on as few heads as
// possible, so a head's 1.25 MB of weights stays resident in that XCD's 4 MiB L2.
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
    float* ube;            // [B] or null
    const float* novelty;  // [B] produced by hash_kernel before the forward (null -> every state novel... see api)
};

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kFcM = 64;    // batch rows per workgroup
constexpr int kFcKC = 32;   // K chunk
constexpr int kFcLD = 68;   // LDS leading dimension (floats) of the K-major tiles: 64 rows + 4 pad
constexpr int kFcH = 256;   // hidden width the kernel is specialised for (config.py:24)

// stage one K chunk of the observation tile: global uint8 [row][k] -> LDS f32 [k][row]
struct AStage {
    uint32_t lo, hi;  // 8 observation bytes of (row = tid>>2, k = chunk*32 + (tid&3)*8 ..)
};

__device__ __forceinline__ AStage a_stage_load(const uint8_t* obs, int B, int in_dim, int row0, int chunk, int tid, bool fast) {
    AStage s;
    s.lo = 0; s.hi = 0;
    const int row = row0 + (tid >> 2);
    const int k = chunk * kFcKC + (tid & 3) * 8;
    if (row < B) {
        const uint8_t* p = obs + static_cast<long>(row) * in_dim + k;
        if (fast && k + 8 <= in_dim) {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            s.lo = v.x; s.hi = v.y;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t byte = (k + i < in_dim) ? p[i] : 0u;
                if (i < 4) s.lo |= byte << (8 * i); else s.hi |= byte << (8 * (i - 4));
            }
        }
    }
    return s;
}

__device__ __forceinline__ void a_stage_store(float* As, const AStage& s, int tid) {
    const int row = tid >> 2, kk = (tid & 3) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t byte = ((i < 4 ? s.lo : s.hi) >> (8 * (i & 3))) & 0xFFu;
        As[(kk + i) * kFcLD + row] = static_cast<float>(byte);
    }
}

// B operand of one chunk: 8 k-steps x float4 (4 interleaved column tiles) per lane
struct BChunk {
    floatx4 v[kFcKC / 4];
};

__device__ __forceinline__ void b_chunk_load(BChunk& bc, const float* W, int K, int ldw, int k0, int col, int kq) {
#pragma unroll
    for (int i = 0; i < kFcKC / 4; ++i) {
        const int k = k0 + 4 * i + kq;
        if (k < K) bc.v[i] = *reinterpret_cast<const floatx4*>(W + static_cast<long>(k) * ldw + col);
        else bc.v[i] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

// 8 k-steps x 16 MFMAs: acc[s][t] += A[rows 4i+s][k] * B[k][cols 4j+t]
__device__ __forceinline__ void mma_chunk(floatx4 (&acc)[4][4], const float* At /* LDS [kc][LD] */, const BChunk& bc, int r, int kq) {
#pragma unroll
    for (int i = 0; i < kFcKC / 4; ++i) {
        const floatx4 a = *reinterpret_cast<const floatx4*>(At + (4 * i + kq) * kFcLD + 4 * r);
        const floatx4 b = bc.v[i];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t], acc[s][t], 0, 0, 0);
    }
}

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

// head_list: up to 4 head ids (0 value, 1 ube, 2 exploitation, 3 exploration) to evaluate
struct HeadList {
    int n;
    int id[4];
};

__global__ void __launch_bounds__(256, 1) fc_forward_kernel(NetDev net, const uint8_t* __restrict__ obs, int B, NetOut out, HeadList heads) {
    __shared__ __attribute__((aligned(16))) float As[2][kFcKC * kFcLD];
    __shared__ __attribute__((aligned(16))) float H1[kFcH * kFcLD];
    __shared__ __attribute__((aligned(16))) float H2[kFcH * kFcLD];

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
    const int n0 = wave * 64;
    const int K1 = net.in_dim;
    const float *W1 = net.w[head * 3], *W2 = net.w[head * 3 + 1], *W3 = net.w[head * 3 + 2];
    const float *b1 = net.b[head * 3], *b2 = net.b[head * 3 + 1], *b3 = net.b[head * 3 + 2];
    const bool fast = (K1 % 8 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 7) == 0);

    floatx4 acc[4][4];
    BChunk bcur, bnext;

    // ---------------- layer 1: [64 x K1] x [K1 x 256] ----------------
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nchunk1 = (K1 + kFcKC - 1) / kFcKC;
    {
        const AStage a0 = a_stage_load(obs, B, K1, row0, 0, tid, fast);
        b_chunk_load(bcur, W1, K1, kFcH, 0, n0 + 4 * r, kq);
        a_stage_store(As[0], a0, tid);
    }
    __syncthreads();
    for (int c = 0; c < nchunk1; ++c) {
        AStage an;
        const bool more = c + 1 < nchunk1;
        if (more) {
            an = a_stage_load(obs, B, K1, row0, c + 1, tid, fast);
            b_chunk_load(bnext, W1, K1, kFcH, (c + 1) * kFcKC, n0 + 4 * r, kq);
        }
        mma_chunk(acc, As[c & 1], bcur, r, kq);
        if (more) {
            a_stage_store(As[(c + 1) & 1], an, tid);
            bcur = bnext;
        }
        __syncthreads();
    }
    store_hidden(H1, acc, b1, n0, r, kq);
    // prefetch W2 chunk 0 while the tile settles
    b_chunk_load(bcur, W2, kFcH, kFcH, 0, n0 + 4 * r, kq);
    __syncthreads();

    // ---------------- layer 2: [64 x 256] x [256 x 256] ----------------
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int c = 0; c < kFcH / kFcKC; ++c) {
        const bool more = c + 1 < kFcH / kFcKC;
        if (more) b_chunk_load(bnext, W2, kFcH, kFcH, (c + 1) * kFcKC, n0 + 4 * r, kq);
        mma_chunk(acc, H1 + c * kFcKC * kFcLD, bcur, r, kq);
        if (more) bcur = bnext;
    }
    store_hidden(H2, acc, b2, n0, r, kq);
    __syncthreads();

    // ---------------- layer 3 + head activation (VALU, k ascending) ----------------
    const int No = head < 2 ? 1 : net.A;
    const int row = tid & 63;
    const bool row_ok = row0 + row < B;
    for (int o = wave; o < No; o += 4) {
        float z = 0.0f;
#pragma unroll 8
        for (int k = 0; k < kFcH; ++k) z = __builtin_fmaf(H2[k * kFcLD + row], W3[static_cast<long>(k) * No + o], z);
        z = z + b3[o];
        if (!row_ok) continue;
        const long gi = row0 + row;
        if (head == 0) {
            if (out.value) out.value[gi] = det_tanh(z);                       // fully_connected.py:55
        } else if (head == 1) {
            float u = 0.5f * (det_tanh(z) + 1.0f);                            // :64
            u = u * net.max_u;                                                // :93
            const float nov = out.novelty ? out.novelty[gi] : 0.0f;
            u = nov > u ? nov : u;                                            // :95
            u = u < 0.0f ? 0.0f : u;                                          // :96
            u = u > net.max_u ? net.max_u : u;
            if (out.ube) out.ube[gi] = u;
        } else {
            float* dst = head == 2 ? out.exploit : out.explore;               // :68-81
            if (dst) dst[gi * net.A + o] = z;
        }
    }
}

// ---- XXHash-variant novelty (src/network/hashes.py:162-229): 4 consecutive lanes per sample, lane i = quarter i ----
// mode: 0 = lookup (writes novelty/seen/indices), 1 = update (scatter-OR, hashes.py:45-50)
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
        if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) {
            for (; l + 4 <= L; l += 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(p + l);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t lane_val = f2u(static_cast<float>((w >> (8 * j)) & 0xFFu));  // astype(f32) + bitcast :207
                    acc = rotl32(acc + lane_val * P2, 13) * P1;                                 // :176-181
                }
            }
        }
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
