// net_fc.hpp -- EpistemicFullyConnectedAZNet forward (src/network/fully_connected.py:41-101) and the XXHash-variant
// novelty set (src/network/hashes.py:23-50,162-229) for gfx950.
//
// Numerics contract (DESIGN.md "FC forward"): hk.Linear is y = x @ w + b with, per output element,
//     acc = +0;  acc = fmaf(x[k], w[k][j], acc) for k = 0..K-1;  y = acc + b[j]
// which is bit-for-bit what a chain of v_mfma_f32_16x16x4_f32 instructions computes (f32 in, f32 accumulate, one
// rounding per product, k ascending).  Layers 1 and 2 of every head are single chains.  Layer 3 (K = 256 hidden units) is FOUR
// chains, one per quarter of k (the 64 hidden columns a wavefront owns), each from +0, combined as ((c0 + c1) + (c2 + c3)) + b[j]
// (oracle: o_linear_parts): every wavefront runs the chain over the columns it has just produced instead of one wavefront running
// a 64-deep dependent MFMA chain while three wait.  tanh is det_tanh (dmath.hpp).
//
// Two kernels with the same numerics: fc16_forward_kernel (16 rows per workgroup, further down; the default whenever its
// whole grid is resident at once) and fc_forward_kernel (64 rows per workgroup, described here; large batches).
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
    const uint32_t* ztab;  // device [4][hash_len/4 + 1]: accumulator of hash lane i after n all-zero stripes (one-hot fast path)
    const uint32_t* hot_index;  // device [in_dim]: get_indices() of the row that is zero except obs[h] = 1 (hashes.py:162-229 / :124-139)
    const float* sim_matrix;    // non-null: hash_class = SimHash, random matrix [hash_len, hash_bits] (hashes.py:124-139)
    // Subleq-16 one-launch kernel (search_subleq_persist.hpp): layer-1 weights of the four heads in its packed input order (sq16_l1_src),
    // [32 l1q_halves][256] each; l1q_mode -1: not built
    const float* w1q[4];
    int l1q_mode, l1q_halves;
    // Layer-1 input order (eaz_net_set_input_order, a switch of the numerics contract): the chain of layer 1 visits the inputs in the order
    // in_perm[0], in_perm[1], ... instead of ascending.  Row kp of the device copies w[0], w[3], w[6], w[9] then holds the weights of input
    // in_perm[kp] and every dense forward reads its observation rows through the same permutation (fc_permute_obs_kernel); null: ascending.
    const int32_t* in_perm;
    const int32_t* in_inv;    // the inverse: position of input k in the chain
    const float* w1raw[4];    // layer-1 weights of the four heads with their rows in the ORIGINAL input order (= w[3 h] without an order)
};

struct NetOut {
    float* exploit;        // [B,A] or null
    float* explore;        // [B,A] or null
    float* value;          // [B] or null
    float* ube;            // [B] or null: 0.5 * (tanh + 1) * max_u, BEFORE the novelty max / clip (see ube_combine)
};

// rows of a head's W1 in the network's input order: out[kp][c] = raw[perm[kp]][c]   (grid = in_dim, block = H threads)
__global__ void fc_permute_w1_kernel(const float* __restrict__ raw, const int32_t* __restrict__ perm, float* __restrict__ out, int H) {
    const int kp = blockIdx.x;
    for (int c = threadIdx.x; c < H; c += blockDim.x) out[static_cast<long>(kp) * H + c] = raw[static_cast<long>(perm[kp]) * H + c];
}
// observation rows in the network's input order: out[b][kp] = obs[b][perm[kp]]
__global__ void __launch_bounds__(256) fc_permute_obs_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ perm, uint8_t* __restrict__ out,
                                                            int B, int K) {
    const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= static_cast<long>(B) * K) return;
    const int b = static_cast<int>(i / K), kp = static_cast<int>(i - static_cast<long>(b) * K);
    out[i] = obs[static_cast<long>(b) * K + perm[kp]];
}

__global__ void __launch_bounds__(256) ube_combine_kernel(float* ube, const float* novelty, float max_u, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) ube[b] = ube_combine(ube[b], novelty[b], max_u);
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

// __builtin_amdgcn_sched_barrier mask used around the MFMA groups of the ResNet kernel: VALU (0x2), SALU (0x4) and LDS (0x80)
// instructions may move across, vector-memory loads and MFMAs may not -- the weight prefetch stays in front of the MFMAs it
// overlaps with, while address arithmetic and operand reads still fill the MFMA shadow.  (Measured on the FC kernels the same
// pinning LOSES: with 12 waves per CU the compiler's just-in-time loads are hidden by the other waves, and eight loads issued at
// once by every wave only add queueing -- fc16 26.8 -> 32.4 us -- so the FC loops are left to the scheduler.)
constexpr int kSchedPin = 0x2 | 0x4 | 0x80;

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
constexpr int kFcMaskBytes = 8 * 4 * kFcMaxBlocks;  // one 64-bit mask per (256-input block, wave)
constexpr int kFcLdsBytes = 2 * kFcH1Bytes + kFcMaskBytes + kFcH * kFcW3Cols * 4;
static_assert(kFcKC * kFcM <= kFcH1Bytes, "the observation chunk (64 rows x 1024 inputs, uint8) aliases the H2 tile");

// head_list: up to 4 head ids (0 value, 1 ube, 2 exploitation, 3 exploration) to evaluate
struct HeadList {
    int n;
    int id[4];
};

// Observation chunk in flight (1024 inputs x 64 rows = 64 KB): thread tid holds 4 units u = tid + 256*q; unit u covers
// the 16 inputs k = 16*(u & 63) .. +15 of the 4 rows 4*(u >> 6) .. +3, i.e. one 16-byte load per row (16 loads per thread,
// each wave-instruction reads 1 KiB contiguous per row).
struct AStage {
    uint4 w[4][4];  // [q][i]: row 4 * ((tid >> 6) + 4 * q) + i
};

// Loads are unconditional from a clamped in-range address and zeroed by a select afterwards (no divergent control flow).
template <bool FAST>
__device__ __forceinline__ void a_stage_load(AStage& s, const uint8_t* obs, int B, int in_dim, int row0, int chunk, int tid) {
    const int k = chunk * kFcKC + 16 * (tid & 63);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * ((tid >> 6) + 4 * q) + i;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (FAST) {
                // in_dim % 16 == 0 and a 16-byte aligned base: the 16 inputs are either all inside the row or all outside
                const bool ok = row < B && k < in_dim;
                const long off = static_cast<long>(ok ? row : 0) * in_dim + (ok ? k : 0);
                const uint4 ld = *reinterpret_cast<const uint4*>(obs + off);
                v.x = ok ? ld.x : 0u; v.y = ok ? ld.y : 0u; v.z = ok ? ld.z : 0u; v.w = ok ? ld.w : 0u;
            } else {
                uint32_t d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const bool ok = row < B && k + b < in_dim;
                    const long off = ok ? static_cast<long>(row) * in_dim + k + b : 0;
                    const uint32_t byte = obs[off];
                    d[b >> 2] |= (ok ? byte : 0u) << (8 * (b & 3));
                }
                v = make_uint4(d[0], d[1], d[2], d[3]);
            }
            s.w[q][i] = v;
        }
}

// spread the low 16 bits of x to bit positions 0, 4, 8, ..., 60
__device__ __forceinline__ unsigned long long spread4(unsigned long long x) {
    x &= 0xFFFFull;
    x = (x | (x << 24)) & 0x000000FF000000FFull;
    x = (x | (x << 12)) & 0x000F000F000F000Full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}

// Each wave publishes, per 256-input block, the mask of k-steps (4 inputs) in which ITS 16 rows hold a non-zero entry
// (plain stores into its own slot; the block mask is the OR of the four slots).  k-step ks = 4 * (tid & 63) + w of the
// chunk lives in block ks >> 6 at bit ks & 63.
__device__ __forceinline__ void a_stage_masks(unsigned long long* kmask /* [4 blocks][4 waves] */, const AStage& s, int tid) {
    uint32_t any[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) { any[0] |= s.w[q][i].x; any[1] |= s.w[q][i].y; any[2] |= s.w[q][i].z; any[3] |= s.w[q][i].w; }
    unsigned long long bw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) bw[w] = __ballot(any[w] != 0);  // bit per lane (= per group of 16 inputs)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        unsigned long long m = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) m |= spread4(bw[w] >> (16 * kb)) << w;
        if ((tid & 63) == 0) kmask[kb * 4 + (tid >> 6)] = m;
    }
}

__device__ __forceinline__ unsigned long long block_mask(const unsigned long long* kmask, int kb) {
    return kmask[kb * 4] | kmask[kb * 4 + 1] | kmask[kb * 4 + 2] | kmask[kb * 4 + 3];
}

// Observation tile in LDS: ROW-major uint8 [64 rows][1024 inputs] (64 KB), stored with the 16-byte loads as they arrive
// (ds_write_b128, conflict-free because lanes differ in the 16-byte chunk).  The chunk index is XOR-swizzled with the
// row group so that the MFMA operand reads (4 rows x 1 byte per lane, rows 4 apart between lanes) spread over the banks.
__device__ __forceinline__ int a_tile_offset(int row, int k) {
    return row * kFcKC + ((((k >> 4) ^ ((row >> 2) & 7))) << 4) + (k & 15);
}

__device__ __forceinline__ void a_stage_store(uint8_t* At, const AStage& s, int tid) {
    const int k = 16 * (tid & 63);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * ((tid >> 6) + 4 * q) + i;
            *reinterpret_cast<uint4*>(At + a_tile_offset(row, k)) = s.w[q][i];
        }
}

// A operand of one k-step for lane (r, kq): inputs k = 4*ks + kq of rows 4r .. 4r+3 (all four share one row group)
__device__ __forceinline__ floatx4 a_tile_read(const uint8_t* At, int ks, int r, int kq) {
    const int k = 4 * ks + kq;
    const int base = 4 * r * kFcKC + ((((k >> 4) ^ (r & 7))) << 4) + (k & 15);
    return floatx4{static_cast<float>(At[base]), static_cast<float>(At[base + kFcKC]), static_cast<float>(At[base + 2 * kFcKC]),
                   static_cast<float>(At[base + 3 * kFcKC])};
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
    // A group always has kFcG slots: slots past the last real k-step repeat it with an all-zero weight fragment, which
    // leaves the accumulators bit-for-bit unchanged (acc is never -0: it starts at +0 and x + (+-0) keeps x).  This keeps
    // the MFMA body free of per-slot branches (one 128-MFMA body per call site instead of one per group size).
    KGroup g;
    g.n = 0;
    int last = 0;
#pragma unroll
    for (int i = 0; i < kFcG; ++i) {
        if (mask) {
            last = __builtin_ctzll(mask);
            mask &= mask - 1;
            g.n = i + 1;
        }
        g.ks[i] = last;
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
#pragma clang loop unroll(disable)
    while (true) {
        const KGroup gn = next_group(mask);
        if (gn.n) load_group(bn, gn, W, K, k0, col, kq);
#pragma unroll
        for (int i = 0; i < kFcG; ++i) {
            floatx4 a;
            if constexpr (U8A) {
                a = a_tile_read(static_cast<const uint8_t*>(Abase), gc.ks[i], r, kq);
            } else {
                const float* At = static_cast<const float*>(Abase);
                a = *reinterpret_cast<const floatx4*>(At + (4 * gc.ks[i] + kq) * kFcLD + 4 * r);
            }
            EAZ_MFMA16(acc, a, bc.v[i]);
        }
        if (gn.n == 0) break;
        gc = gn;
        bc = bn;
    }
}

#ifdef EAZ_FC_STAMPS
// Diagnostic build only (make diag): shader-clock stamps at the phase boundaries of workgroup `blockIdx.x`, written to a
// buffer nothing else reads.  Never compiled into libeaz.so.
__device__ unsigned long long g_fc_stamps[1024 * 8];
#define EAZ_STAMP(i)                                                                                   \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                   \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            g_fc_stamps[blockIdx.x * 8 + (i)] = t_;                                                    \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#else
#define EAZ_STAMP(i)
#endif

// FAST: in_dim % 16 == 0 and a 16-byte aligned observation batch (every configuration of the reference); the generic
// instantiation stages the tile with guarded byte loads.
template <bool FAST>
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

    EAZ_STAMP(0);
    floatx4 acc[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---------------- layer 1: [64 x K1] x [K1 x 256], all-zero k-steps skipped ----------------
    const int nchunk1 = (K1 + kFcKC - 1) / kFcKC;
    unsigned long long executed = 0;
    // layer-3 weights: [256][No] floats, staged now (coalesced) and read back as LDS broadcasts after layer 2
    const int No = head < 2 ? 1 : net.A;
    float* W3s = reinterpret_cast<float*>(lds + 2 * kFcH1Bytes + kFcMaskBytes);
    const bool w3_lds = No <= kFcW3Cols;
    AStage st;
    a_stage_load<FAST>(st, obs, B, K1, row0, 0, tid);
    if (w3_lds) {
        // thread k stages row k of W3, zero-padded to 16 columns (B operand of the layer-3 MFMA): all loads in flight
        // together with the observation loads above, one round trip
        float wv[kFcW3Cols];
#pragma unroll
        for (int j = 0; j < kFcW3Cols; ++j) wv[j] = j < No ? W3[tid * No + j] : 0.0f;
#pragma unroll
        for (int j = 0; j < kFcW3Cols; j += 4)
            *reinterpret_cast<floatx4*>(W3s + tid * kFcW3Cols + j) = floatx4{wv[j], wv[j + 1], wv[j + 2], wv[j + 3]};
    }
    for (int c = 0; c < nchunk1; ++c) {
        a_stage_masks(&kmask[16 * c], st, tid);
        a_stage_store(R2, st, tid);
        __syncthreads();  // chunk c and its masks are visible
        EAZ_STAMP(1);
        if (c + 1 < nchunk1) a_stage_load<FAST>(st, obs, B, K1, row0, c + 1, tid);  // in flight during the MFMAs below
#pragma unroll 1
        for (int kb = 0; kb < 4; ++kb) {
            const unsigned long long mv = block_mask(&kmask[16 * c], kb);
            // the mask is the same in every lane: move it to scalar registers so the k-step loop is scalar control flow
            const unsigned long long m = static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(mv))) |
                                         (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(mv >> 32))) << 32);
            executed += __builtin_popcountll(m);
            mma_block<true>(acc, R2 + kb * 256, m, W1, K1, c * kFcKC + kb * 256, col, r, kq);
        }
        if (c + 1 < nchunk1) __syncthreads();  // everyone is done reading chunk c before it is overwritten
    }
    EAZ_STAMP(2);
    store_hidden(H1, acc, b1, n0, r, kq);
    __syncthreads();  // H1 complete; region 2 (observation chunk) is dead from here on
    EAZ_STAMP(3);

    // ---------------- layer 2: [64 x 256] x [256 x 256], dense ----------------
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    mma_block<false>(acc, H1, ~0ull, W2, kFcH, 0, col, r, kq);
    EAZ_STAMP(4);
    store_hidden(H2, acc, b2, n0, r, kq);
    __syncthreads();
    EAZ_STAMP(5);

    // ---------------- layer 3 + head activation ----------------
    // out <= 16: one 16x16x4 MFMA chain per wave (rows 16*wave..+15, W3 zero-padded to 16 columns in LDS);
    // out == 256: the layer-2 code path; anything else: VALU fmaf chains.  All three are k-ascending fmaf chains.
    auto emit = [&](int row, int o, float z) {
        if (row0 + row >= B) return;
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
    };
    if (w3_lds) {
        // four independent chains of 16 k-steps (quarters of k), then ((c0 + c1) + (c2 + c3))
        floatx4 aq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) aq[q] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
        for (int ks = 0; ks < kFcH / 16; ++ks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * (16 * q + ks) + kq;
                aq[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(H2[k * kFcLD + 16 * wave + r], W3s[k * kFcW3Cols + r], aq[q], 0, 0, 0);
            }
        }
        const floatx4 a3 = (aq[0] + aq[1]) + (aq[2] + aq[3]);
        if (r < No) {
            const float bo = b3[r];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) emit(16 * wave + 4 * kq + reg, r, a3[reg] + bo);
        }
    } else if (No == kFcH) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        {
            // quarters of k as separate chains: (c0 + c1) + (c2 + c3)
            floatx4 c1[4][4], t01[4][4];
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                floatx4 (&dst)[4][4] = (q & 1) ? c1 : acc;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t) dst[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                mma_block<false>(dst, H2, 0xFFFFull << (16 * q), W3, kFcH, 0, col, r, kq);
                if (q & 1) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const floatx4 sum = acc[s][t] + c1[s][t];
                            if (q == 1) t01[s][t] = sum; else acc[s][t] = t01[s][t] + sum;
                        }
                }
            }
        }
        // only policy heads can have 256 outputs: plain stores (keeps the 64 epilogue sites free of the tanh code)
        float* dst = head == 2 ? out.exploit : out.explore;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float bo = b3[col + t];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int row = 16 * kq + 4 * reg + s;
                    if (dst && row0 + row < B) dst[static_cast<long>(row0 + row) * net.A + col + t] = acc[s][t][reg] + bo;
                }
        }
    } else {
        const int row = tid & 63;
        for (int o = wave; o < No; o += 4) {
            float zq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float z = 0.0f;
#pragma unroll 16
                for (int k = q * (kFcH / 4); k < (q + 1) * (kFcH / 4); ++k) z = __builtin_fmaf(H2[k * kFcLD + row], W3[static_cast<long>(k) * No + o], z);
                zq[q] = z;
            }
            emit(row, o, ((zq[0] + zq[1]) + (zq[2] + zq[3])) + b3[o]);
        }
    }
    EAZ_STAMP(6);
    if (kstep_counter && tid == 0) atomicAdd(kstep_counter, 4 * executed);  // unit: one k-step of a 16-row tile
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

// ---- SimHash (src/network/hashes.py:100-139): index = sum_j (product_j < 0) << j with product = x @ random_matrix [n, bits] ------------
// 32 consecutive lanes per sample, lane j = hash bit j: product_j as the hk.Linear chain (acc = +0; acc = fmaf(x[k], R[k][j], acc), k
// ascending; zero inputs are skipped, which is exact).  x[k] is one broadcast byte load per group, R[k][.] one coalesced row.
// mode as hash_kernel.
__global__ void __launch_bounds__(256) simhash_kernel(NetDev net, const uint8_t* __restrict__ obs, int B, float* novelty,
                                                      uint8_t* seen_out, uint32_t* idx_out, int mode) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gtid >> 5, j = gtid & 31;
    const bool live = b < B;
    const int bits = net.hash_bits, n = net.hash_len;
    const uint8_t* p = obs + static_cast<long>(live ? b : 0) * net.in_dim + net.hash_off;
    const float* R = net.sim_matrix + (j < bits ? j : 0);
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) {
        const uint8_t xv = p[k];
        if (xv) acc = __builtin_fmaf(static_cast<float>(xv), R[static_cast<long>(k) * bits], acc);
    }
    const unsigned long long bal = __ballot(live && j < bits && acc < 0.0f);
    const uint32_t index = static_cast<uint32_t>(bal >> (threadIdx.x & 32));  // the 32 lanes of this sample
    if (!live || j != 0) return;
    if (mode == 1) {
        atomicOr(reinterpret_cast<uint32_t*>(net.bitset) + (index >> 5), 1u << (index & 31u));
        return;
    }
    const bool seen = net.bitset ? ((net.bitset[index >> 3] >> (index & 7u)) & 1u) != 0 : false;
    if (novelty) novelty[b] = (seen ? 0.0f : 1.0f) * net.nov_scale;
    if (seen_out) seen_out[b] = seen ? 1 : 0;
    if (idx_out) idx_out[b] = index;
}

// ==================================================================================================================
// fc16_forward_kernel -- the same three layers with 16 batch rows per workgroup.
//
// Why: at B = 4096 the 64-row kernel has 64 x heads workgroups of ~33 k MFMA cycles each on 256 CUs (one per CU, 25 % of
// the CUs idle, nothing to overlap the staging / store phases with).  16-row tiles give 256 x heads workgroups of a quarter
// of the work, three or four co-resident per CU (36 KB of LDS, <= 128 VGPRs): the MFMA pipes of every CU are shared by
// waves that are in different phases, and the critical path of one launch is a 16-row tile, not a 64-row one.
// Each wave owns 64 output columns x 16 rows = 1x4 MFMA tiles; operands per k-step: one LDS read (A) + one 16-byte weight
// load (B) for 4 MFMAs.  Weights are re-read from L2 four times as often as by the 64-row kernel (256 KB of W2 per 16 rows),
// so the host picks this kernel only while every workgroup of the launch is resident at once.
//
// SRC 0: dense uint8 observations, 16-byte loads (in_dim % 16 == 0, aligned base);  SRC 1: dense, guarded byte loads;
// SRC 2: ONE-HOT observations given as the flat index of the single 1 (hot[b]): layer 1 is the row W1[hot] (exactly
//        fmaf(1, w, +0) per column; every other term of the chain is fmaf(0, w, acc) = acc), no observation tile at all,
//        and the novelty lookup of the same observations runs in extra workgroups of the same launch (hash of a one-hot row:
//        precomputed zero-run prefix + the single non-zero round + the remaining zero rounds).
// ==================================================================================================================
constexpr int kF16M = 16;
constexpr int kF16HFloats = kFcH * 16 + (kFcH / 4) * 4;  // K-major [256][16] with a 16-byte skew every 4 k: conflict-free b128 stores
constexpr int kF16HBytes = kF16HFloats * 4;              // 17 408
// + layer-3 weights of the narrow heads [256][w3ld], w3ld = actions rounded up to 4 (4 KB for <= 4 actions: 40 KB in all, four
// workgroups per CU -- the novelty workgroups of the one-hot launch then run beside the three forward workgroups of a CU)
constexpr int kF16LdsBase = 2 * kF16HBytes + kFcMaskBytes;
__host__ __device__ inline int f16_w3ld(int A) { return A <= kFcW3Cols ? ((A + 3) & ~3) : 0; }
__host__ __device__ inline int f16_lds_bytes(int A) { return kF16LdsBase + kFcH * f16_w3ld(A) * 4; }
static_assert(kF16M * kFcKC <= kF16HBytes, "the observation chunk (16 rows x 1024 inputs, uint8) aliases the H2 tile");

__device__ __forceinline__ int h16_off(int k, int row) { return k * 16 + (k >> 2) * 4 + row; }

struct FcIn {
    const uint8_t* obs;     // SRC 0/1: [B, in_dim]
    const int32_t* hot;     // SRC 2: [B] flat index of the single 1
    float* novelty;         // SRC 2: [B] or null
    const uint32_t* ztab;   // SRC 2: [4][L+1] accumulator of hash lane i after n all-zero stripes (L = hash_len / 4)
};

struct A16 {
    uint4 w[4];  // rows 4 * wave + i, inputs 16 * (tid & 63) .. +15 of the chunk
};

template <int SRC>
__device__ __forceinline__ void a16_load(A16& s, const uint8_t* obs, int B, int in_dim, int row0, int chunk, int tid) {
    const int k = chunk * kFcKC + 16 * (tid & 63);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * (tid >> 6) + i;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (SRC == 0) {
            const bool ok = row < B && k < in_dim;
            const long off = static_cast<long>(ok ? row : 0) * in_dim + (ok ? k : 0);
            const uint4 ld = *reinterpret_cast<const uint4*>(obs + off);
            v.x = ok ? ld.x : 0u; v.y = ok ? ld.y : 0u; v.z = ok ? ld.z : 0u; v.w = ok ? ld.w : 0u;
        } else {
            uint32_t d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const bool ok = row < B && k + b < in_dim;
                const long off = ok ? static_cast<long>(row) * in_dim + k + b : 0;
                const uint32_t byte = obs[off];
                d[b >> 2] |= (ok ? byte : 0u) << (8 * (b & 3));
            }
            v = make_uint4(d[0], d[1], d[2], d[3]);
        }
        s.w[i] = v;
    }
}

__device__ __forceinline__ void a16_masks(unsigned long long* kmask /* [4 blocks][4 waves] */, const A16& s, int tid) {
    uint32_t any[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) { any[0] |= s.w[i].x; any[1] |= s.w[i].y; any[2] |= s.w[i].z; any[3] |= s.w[i].w; }
    unsigned long long bw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) bw[w] = __ballot(any[w] != 0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        unsigned long long m = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) m |= spread4(bw[w] >> (16 * kb)) << w;
        if ((tid & 63) == 0) kmask[kb * 4 + (tid >> 6)] = m;
    }
}

// observation tile: row-major uint8 [16][1024]; the 16-byte slot index is XOR-ed with the row (low 4 bits), so the 16 rows
// that one A-operand read touches (same k, rows 0..15) fall into 16 different 16-byte bank groups
__device__ __forceinline__ void a16_store(uint8_t* At, const A16& s, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * (tid >> 6) + i;
        *reinterpret_cast<uint4*>(At + row * kFcKC + ((((tid & 63) ^ row)) << 4)) = s.w[i];
    }
}

// U8A: A = observation tile block (256 inputs at Abase, layer 1); else A = hidden tile (layer 2 / 3)
template <bool U8A>
__device__ __forceinline__ void mma16_block(floatx4 (&acc)[4], const void* Abase, unsigned long long mask, const float* W, int K, int k0,
                                            int col, int r, int kq) {
    KGroup gc = next_group(mask);
    if (gc.n == 0) return;
    BGroup bc, bn;
    load_group(bc, gc, W, K, k0, col, kq);
#pragma clang loop unroll(disable)
    while (true) {
        const KGroup gn = next_group(mask);
        if (gn.n) load_group(bn, gn, W, K, k0, col, kq);
#pragma unroll
        for (int i = 0; i < kFcG; ++i) {
            float a;
            if constexpr (U8A) {
                const int k = 4 * gc.ks[i] + kq;
                a = static_cast<float>(static_cast<const uint8_t*>(Abase)[r * kFcKC + ((((k >> 4) ^ r)) << 4) + (k & 15)]);
            } else {
                a = static_cast<const float*>(Abase)[h16_off(4 * gc.ks[i] + kq, r)];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc.v[i][t], acc[t], 0, 0, 0);
        }
        if (gn.n == 0) break;
        gc = gn;
        bc = bn;
    }
}

// One whole 256-input block of layer 1 without the mask machinery: all 64 k-steps, compile-time trip count (the compiler unrolls
// and schedules it like layer 2).  Used when most k-steps of the block are non-zero anyway: executing an all-zero k-step is
// exact (fmaf(0, w, acc) == acc), and k-steps beyond in_dim read a clamped (valid) weight row against observation bytes that
// the staging zero-filled.
__device__ __forceinline__ void mma16_dense_u8(floatx4 (&acc)[4], const uint8_t* At, const float* W, int K, int k0, int col, int r, int kq) {
#pragma unroll 8
    for (int ks = 0; ks < 64; ++ks) {
        const int kl = 4 * ks + kq, k = k0 + kl;
        const floatx4 b = *reinterpret_cast<const floatx4*>(W + static_cast<long>(k < K ? k : K - 1) * kFcH + col);
        const float a = static_cast<float>(At[r * kFcKC + ((((kl >> 4) ^ r)) << 4) + (kl & 15)]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
    }
}

// accumulators -> relu(acc + bias) -> hidden tile: lane (r, kq) holds rows 4*kq .. 4*kq+3 of column n0 + 4*r + t
__device__ __forceinline__ void store_hidden16(float* Ht, const floatx4 (&acc)[4], const float* bias, int n0, int r, int kq) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = n0 + 4 * r + t;
        const float bj = bias[col];
        floatx4 o;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float v = acc[t][reg] + bj;
            o[reg] = v > 0.0f ? v : 0.0f;
        }
        *reinterpret_cast<floatx4*>(Ht + h16_off(col, 4 * kq)) = o;
    }
}

__device__ __forceinline__ uint32_t xxh_zero_round(uint32_t acc) { return rotl32(acc, 13) * 0x9E3779B1u; }

template <int SRC>
__global__ void __launch_bounds__(256, 3) fc16_forward_kernel(NetDev net, FcIn in, int B, NetOut out, HeadList heads, int per_xcd, int n_fc_blocks,
                                                              unsigned long long* kstep_counter) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    if constexpr (SRC == 2) {
        if (static_cast<int>(blockIdx.x) >= n_fc_blocks) {
            // ---- novelty lookup of one-hot observations (hashes.py:23-38 with the index of hashes.py:162-229 on a row that is zero except
            // obs[hot] = 1) ----
            const int b = (blockIdx.x - n_fc_blocks) * 256 + tid;
            if (b >= B || !in.novelty) return;
            int h = in.hot[b];
            h = h < 0 ? 0 : (h >= net.in_dim ? net.in_dim - 1 : h);
            const uint32_t index = net.hot_index[h];  // tabulated once per network: the index depends on `hot` only
            const bool seen = net.bitset ? ((net.bitset[index >> 3] >> (index & 7u)) & 1u) != 0 : false;
            in.novelty[b] = (seen ? 0.0f : 1.0f) * net.nov_scale;
            return;
        }
    }
    float* H1 = reinterpret_cast<float*>(lds);
    uint8_t* R2 = lds + kF16HBytes;
    float* H2 = reinterpret_cast<float*>(R2);
    unsigned long long* kmask = reinterpret_cast<unsigned long long*>(lds + 2 * kF16HBytes);

    // blockIdx -> (head, row tile): XCD x (= blockIdx % 8) owns row tiles x*per_xcd .. +per_xcd-1 of EVERY head (3-4 heads of
    // weights = 3.8-5 MB against its 4 MiB L2); consecutive blocks of an XCD walk the tiles of one head
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int hsel = j / per_xcd;
    const int tile = xcd * per_xcd + (j - hsel * per_xcd);
    const int n_tiles = (B + kF16M - 1) / kF16M;
    if (hsel >= heads.n || tile >= n_tiles) return;
    const int head = heads.id[hsel];
    const int row0 = tile * kF16M;

    const int wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int n0 = wave * 64, col = n0 + 4 * r;
    const int K1 = net.in_dim;
    const float *W1 = net.w[head * 3], *W2 = net.w[head * 3 + 1], *W3 = net.w[head * 3 + 2];
    const float *b1 = net.b[head * 3], *b2 = net.b[head * 3 + 1], *b3 = net.b[head * 3 + 2];
    const int No = head < 2 ? 1 : net.A;

    EAZ_STAMP(0);
    floatx4 acc[4];
    unsigned long long executed = 0;
    // layer-3 weights of the narrow heads -> LDS [256][16] (zero beyond No), requested first so that the loads fly under layer 1;
    // thread k stages row k.  (Preloading them into the registers of the layer-3 wave after layer 2 put 64 global loads in front
    // of the H2 barrier: 8 k cycles per workgroup.)
    float* W3s = reinterpret_cast<float*>(lds + kF16LdsBase);
    const int w3ld = f16_w3ld(net.A);  // every head of the launch uses the stride of the widest narrow head
    const bool narrow = net.A <= kFcW3Cols;
    if (narrow) {
        for (int c = 0; c < w3ld; c += 4) {
            floatx4 wv;
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[i] = c + i < No ? W3[tid * No + c + i] : 0.0f;
            *reinterpret_cast<floatx4*>(W3s + tid * w3ld + c) = wv;
        }
    }
    if constexpr (SRC == 2) {
        // ---------------- layer 1 of a one-hot row: relu(W1[hot] + b1), thread = output column ----------------
        float w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = row0 + i;
            int h = in.hot[row < B ? row : B - 1];
            h = h < 0 ? 0 : (h >= K1 ? K1 - 1 : h);
            w[i] = W1[static_cast<long>(h) * kFcH + tid];
        }
        const float bj = b1[tid];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            floatx4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = __builtin_fmaf(1.0f, w[4 * i4 + i], 0.0f) + bj;
                o[i] = v > 0.0f ? v : 0.0f;
            }
            *reinterpret_cast<floatx4*>(H1 + h16_off(tid, 4 * i4)) = o;
        }
    } else {
        // ---------------- layer 1: [16 x K1] x [K1 x 256], all-zero k-steps skipped ----------------
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        const int nchunk1 = (K1 + kFcKC - 1) / kFcKC;
        A16 st;
        a16_load<SRC>(st, in.obs, B, K1, row0, 0, tid);
        for (int c = 0; c < nchunk1; ++c) {
            a16_masks(&kmask[16 * c], st, tid);
            a16_store(R2, st, tid);
            __syncthreads();  // chunk c and its masks are visible
            if (c + 1 < nchunk1) a16_load<SRC>(st, in.obs, B, K1, row0, c + 1, tid);
#pragma unroll 1
            for (int kb = 0; kb < 4; ++kb) {
                const unsigned long long mv = block_mask(&kmask[16 * c], kb);
                const unsigned long long m = static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(mv))) |
                                             (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(mv >> 32))) << 32);
                const int k0 = c * kFcKC + kb * 256;
                const int valid = (K1 - k0 + 3) / 4;  // k-steps of this block that lie inside the observation
                if (__builtin_popcountll(m) >= 40 || (valid > 0 && valid < 64 && 4 * __builtin_popcountll(m) >= 3 * valid)) {
                    executed += 64;
                    mma16_dense_u8(acc, R2 + kb * 256, W1, K1, k0, col, r, kq);
                } else {
                    executed += __builtin_popcountll(m);
                    mma16_block<true>(acc, R2 + kb * 256, m, W1, K1, k0, col, r, kq);
                }
            }
            if (c + 1 < nchunk1) __syncthreads();  // everyone is done reading chunk c before it is overwritten
        }
        store_hidden16(H1, acc, b1, n0, r, kq);
    }
    __syncthreads();  // H1 complete; region 2 (observation chunk) is dead from here on
    EAZ_STAMP(1);

    // ---------------- layer 2: [16 x 256] x [256 x 256], dense ----------------
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    mma16_block<false>(acc, H1, ~0ull, W2, kFcH, 0, col, r, kq);
    EAZ_STAMP(2);
    store_hidden16(H2, acc, b2, n0, r, kq);
    // narrow heads read back only the columns (and layer-3 weight rows) their own wave wrote: DS operations of a wave execute in order
    if (net.A <= kFcW3Cols) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
    EAZ_STAMP(3);

    // ---------------- layer 3 + head activation ----------------
    auto emit = [&](int row, int o, float z) {
        if (row0 + row >= B) return;
        const long gi = row0 + row;
        if (head == 0) {
            if (out.value) out.value[gi] = det_tanh(z);                       // fully_connected.py:55
        } else if (head == 1) {
            float u = 0.5f * (det_tanh(z) + 1.0f);                            // :64
            u = u * net.max_u;                                                // :93 (:95-96 joined in ube_combine())
            if (out.ube) out.ube[gi] = u;
        } else {
            float* dst = head == 2 ? out.exploit : out.explore;               // :68-81
            if (dst) dst[gi * net.A + o] = z;
        }
    };
    if (narrow) {
        // every wave: the chain over the 64 hidden columns it owns (quarter `wave` of k), 16 dependent MFMAs; the four partial sums
        // meet in LDS and wave 0 adds them as ((c0 + c1) + (c2 + c3)) + b
        // partial sums [16 rows][w3ld] of wave w: in the H2 columns that wave owns (dead once its chain operands are read; 1088 floats apart)
        float* part = H2;
        constexpr int kPartStride = 64 * 16 + 16 * 4;  // h16_off(64 w, 0)
        floatx4 a3 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        // lanes beyond the staged columns read column 0: their output columns are never emitted (r < No below)
        const float* w3p = W3s + (n0 + kq) * w3ld + (r < w3ld ? r : 0);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(H2[h16_off(n0 + 4 * ks + kq, r)], w3p[(4 * ks) * w3ld], a3, 0, 0, 0);
        if (r < w3ld) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) part[wave * kPartStride + (4 * kq + reg) * w3ld + r] = a3[reg];
        }
        __syncthreads();
        if (wave != 0) return;
        if (r < No) {
            const float bo = b3[r];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 4 * kq + reg;
                const float c0 = part[row * w3ld + r], c1 = part[kPartStride + row * w3ld + r];
                const float c2 = part[2 * kPartStride + row * w3ld + r], c3 = part[3 * kPartStride + row * w3ld + r];
                emit(row, r, ((c0 + c1) + (c2 + c3)) + bo);
            }
        }
        EAZ_STAMP(4);
    } else if (No == kFcH) {
        {
            // quarters of k as separate chains: (c0 + c1) + (c2 + c3)
            floatx4 c1[4], t01[4];
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                floatx4 (&dst)[4] = (q & 1) ? c1 : acc;
#pragma unroll
                for (int t = 0; t < 4; ++t) dst[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                mma16_block<false>(dst, H2, 0xFFFFull << (16 * q), W3, kFcH, 0, col, r, kq);
                if (q & 1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const floatx4 sum = acc[t] + c1[t];
                        if (q == 1) t01[t] = sum; else acc[t] = t01[t] + sum;
                    }
                }
            }
        }
        float* dst = head == 2 ? out.exploit : out.explore;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float bo = b3[col + t];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 4 * kq + reg;
                if (dst && row0 + row < B) dst[static_cast<long>(row0 + row) * net.A + col + t] = acc[t][reg] + bo;
            }
        }
    } else {
        const int row = tid & 15;
        for (int o = tid >> 4; o < No; o += 16) {
            float zq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float z = 0.0f;
#pragma unroll 16
                for (int k = q * (kFcH / 4); k < (q + 1) * (kFcH / 4); ++k) z = __builtin_fmaf(H2[h16_off(k, row)], W3[static_cast<long>(k) * No + o], z);
                zq[q] = z;
            }
            emit(row, o, ((zq[0] + zq[1]) + (zq[2] + zq[3])) + b3[o]);
        }
    }
    // narrow heads: only wave 0 gets here; otherwise wave 0 reports for the workgroup
    if (kstep_counter && lane == 0 && wave == 0 && executed) atomicAdd(kstep_counter, executed);
}

// ---- any linear_layer_size other than 256 (config.py:24): plain fmaf chains, not tuned -----------------------------------------------------
// The MFMA kernels above are specialised for H = 256 (tile shapes, LDS layouts, the four 64-column quarters of layer 3).  This kernel is the
// numerics contract written out: thread = output column, acc = +0; acc = fmaf(x[k], w[k][j], acc) for k ascending; + b; relu -- and the
// third layer as four contiguous quarter chains added ((c0 + c1) + (c2 + c3)) when H % 4 == 0, one chain otherwise (oracle o_linear_parts).
// One workgroup = kFgRows rows of one head; hidden activations in LDS ([rows][H] twice).
constexpr int kFgRows = 4;
__host__ __device__ inline int fg_lds_bytes(int H) { return 2 * kFgRows * H * static_cast<int>(sizeof(float)); }
__global__ void __launch_bounds__(256) fc_generic_forward_kernel(NetDev net, const uint8_t* __restrict__ obs, int B, NetOut out, HeadList heads) {
    extern __shared__ __attribute__((aligned(16))) float fg_sm[];
    const int H = net.H, K1 = net.in_dim, tid = threadIdx.x;
    const int head = heads.id[blockIdx.y];
    const int row0 = blockIdx.x * kFgRows;
    float* h1 = fg_sm;
    float* h2 = fg_sm + kFgRows * H;
    const float* W1 = net.w[head * 3], *W2 = net.w[head * 3 + 1], *W3 = net.w[head * 3 + 2];
    const float* b1 = net.b[head * 3], *b2 = net.b[head * 3 + 1], *b3 = net.b[head * 3 + 2];
    for (int j = tid; j < H; j += 256) {  // layer 1 (fully_connected.py:45-47: observation bytes as float32)
        float acc[kFgRows];
#pragma unroll
        for (int r = 0; r < kFgRows; ++r) acc[r] = 0.0f;
        for (int k = 0; k < K1; ++k) {
            const float w = W1[static_cast<long>(k) * H + j];
#pragma unroll
            for (int r = 0; r < kFgRows; ++r) {
                const int row = row0 + r < B ? row0 + r : B - 1;
                acc[r] = __builtin_fmaf(static_cast<float>(obs[static_cast<long>(row) * K1 + k]), w, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < kFgRows; ++r) { const float v = acc[r] + b1[j]; h1[r * H + j] = v > 0.0f ? v : 0.0f; }
    }
    __syncthreads();
    for (int j = tid; j < H; j += 256) {  // layer 2
        float acc[kFgRows];
#pragma unroll
        for (int r = 0; r < kFgRows; ++r) acc[r] = 0.0f;
        for (int k = 0; k < H; ++k) {
            const float w = W2[static_cast<long>(k) * H + j];
#pragma unroll
            for (int r = 0; r < kFgRows; ++r) acc[r] = __builtin_fmaf(h1[r * H + k], w, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < kFgRows; ++r) { const float v = acc[r] + b2[j]; h2[r * H + j] = v > 0.0f ? v : 0.0f; }
    }
    __syncthreads();
    const int No = head < 2 ? 1 : net.A;
    const int parts = (H % 4 == 0) ? 4 : 1, Kp = H / parts;
    for (int idx = tid; idx < kFgRows * No; idx += 256) {  // layer 3 + the head's output (fully_connected.py:55,64,68-81,93)
        const int r = idx / No, o = idx - r * No;
        float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int q = 0; q < parts; ++q) {
            float z = 0.0f;
            for (int k = q * Kp; k < (q + 1) * Kp; ++k) z = __builtin_fmaf(h2[r * H + k], W3[static_cast<long>(k) * No + o], z);
            c[q] = z;
        }
        float z = parts == 4 ? (c[0] + c[1]) + (c[2] + c[3]) : c[0];
        z = z + b3[o];
        if (row0 + r >= B) continue;
        const long gi = row0 + r;
        if (head == 0) { if (out.value) out.value[gi] = det_tanh(z); }
        else if (head == 1) {
            float u = 0.5f * (det_tanh(z) + 1.0f);
            u = u * net.max_u;  // :93 (:95-96 joined in ube_combine())
            if (out.ube) out.ube[gi] = u;
        } else {
            float* dst = head == 2 ? out.exploit : out.explore;
            if (dst) dst[gi * net.A + o] = z;
        }
    }
}

}  // namespace eaz
