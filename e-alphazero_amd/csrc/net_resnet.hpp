// net_resnet.hpp -- EpistemicResidualAZNet forward (src/network/resnet.py:42-135; BlockV2 :26-39) for gfx950.
//
// Numerics contract (DESIGN.md "ResNet forward"), identical to the CPU oracle:
//   hk.Conv2D 3x3 SAME : acc = +0; for tap (kh, kw) row-major, ci ascending: acc = fmaf(x[nb][ci], w[tap][ci][co], acc),
//                        out-of-board taps contribute an exact 0; y = acc + b[co]          -> one f32 MFMA chain, K = 9*Cin
//   hk.BatchNorm (eval): y = (x - mean) * inv + offset, inv = scale / sqrt(var + 1e-5) precomputed on the host
//   hk.Linear / 1x1 conv: k-ascending fmaf chain, then + b
//
// Kernel shape: one 256-thread workgroup carries THREE samples (3 x 42 = 126 board cells = rows of the implicit GEMM, padded to
// 128) through the WHOLE tower without touching HBM: the residual stream X and the activated conv input U live in LDS
// ([128][65] f32 each + one zero row, 67 KB -> two workgroups per CU, so one workgroup's MFMAs fill the other's barriers and
// epilogues), weights (147 KB per 3x3 conv, shared by every workgroup) stream from L2 one group of 8 k-steps ahead.  Each wave
// owns 32 rows x all 64 output channels = 2x4 tiles of v_mfma_f32_16x16x4_f32; the A operand is gathered from U through
// per-tap neighbour-row pointers (2 ds_read_b32 per k-step), so no im2col buffer exists.  (Six samples / 4x4 tiles per wave, one
// workgroup per CU, measured 2.87 ms at B = 8192: 81 % MFMA issue in the tower, 11 % lost to the last partial round of 1366
// workgroups on 256 CUs.)
#pragma once
#include "net_fc.hpp"

namespace eaz {

constexpr int kRnSamples = 3;   // samples per workgroup
constexpr int kRnRows = 128;    // GEMM rows per workgroup (126 used)
constexpr int kRnMT = kRnRows / 64;  // 16-row MFMA tiles per wave
constexpr int kRnLD = 65;       // LDS leading dimension (floats)
constexpr int kRnC = 64;        // channels the kernel is specialised for (resnet.py:48)
constexpr int kRnMaxConv = 33;  // conv0 + 2 * 16 blocks
constexpr int kRnZeroRow = 2 * kRnRows;  // one all-zero row behind the X and U tiles: the source of every off-board tap
constexpr int kRnLdsBytes = (2 * kRnRows + 1) * kRnLD * 4;

struct ResNetDev {
    int H, W, Cin, cin_log2, nb, A, P;
    const float* p;  // packed parameters (device)
    int conv_w[kRnMaxConv], conv_b[kRnMaxConv], conv_k[kRnMaxConv];  // conv 0 = input conv, then (conv1, conv2) per block; K padded to 4
    int bn_mean[kRnMaxConv], bn_inv[kRnMaxConv], bn_off[kRnMaxConv];  // bn1 / bn2 per block at 2k, 2k+1; final bn at 2*nb
    int hc_w[4], hc_b[4], hbn_mean[4], hbn_inv[4], hbn_off[4];        // heads: policy, exploration, value, ube
    int lin_w[6], lin_b[6];                                            // policy, exploration, value fc1, value fc2, ube fc1, ube fc2
};

// rows of this lane: m_s = 16*MT*wave + MT*i + s  (i = lane & 15, s < MT); cell (h, w) of the row and a 9-bit mask of the taps
// that stay on the board
struct RnLane {
    int m[kRnMT];
    unsigned tapmask[kRnMT];
};

__device__ __forceinline__ RnLane rn_lane(const ResNetDev& n, int wave, int lane, int rows_valid) {
    RnLane L;
#pragma unroll
    for (int s = 0; s < kRnMT; ++s) {
        const int m = 16 * kRnMT * wave + kRnMT * (lane & 15) + s;
        L.m[s] = m;
        unsigned mask = 0;
        if (m < rows_valid) {
            const int cell = m % n.P, h = cell / n.W, w = cell - h * n.W;
            for (int t = 0; t < 9; ++t) {
                const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
                if (hh >= 0 && hh < n.H && ww >= 0 && ww < n.W) mask |= 1u << t;
            }
        }
        L.tapmask[s] = mask;
    }
    return L;
}

// one 3x3 (or input) convolution as an implicit GEMM: acc[s][t] += gather(U)[rows][k] * Wc[k][cols]
typedef float rn_avec __attribute__((ext_vector_type(kRnMT)));
#define EAZ_RN_MFMA(acc, a, b)                                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < kRnMT; ++s_) _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) \
        acc[s_][t_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], b[t_], acc[s_][t_], 0, 0, 0)

__device__ __forceinline__ void rn_conv(floatx4 (&acc)[kRnMT][4], const ResNetDev& n, const float* U, const float* Wc, int K, int cin_log2,
                                        const RnLane& L, int lane) {
    const int r = lane & 15, kq = lane >> 4, col = 4 * r;
    const int nsteps = K >> 2;
    BGroup bc, bn;
    auto load_b = [&](BGroup& b, int ks0) {
#pragma unroll
        for (int i = 0; i < kFcG; ++i) {
            const int k = 4 * (ks0 + i) + kq;
            const bool ok = ks0 + i < nsteps;
            const floatx4 v = *reinterpret_cast<const floatx4*>(Wc + static_cast<long>(ok ? k : kq) * kRnC + col);
            b.v[i] = floatx4{ok ? v[0] : 0.0f, ok ? v[1] : 0.0f, ok ? v[2] : 0.0f, ok ? v[3] : 0.0f};
        }
    };
    load_b(bc, 0);
#pragma clang loop unroll(disable)
    for (int ks0 = 0; ks0 < nsteps; ks0 += kFcG) {
        load_b(bn, ks0 + kFcG);  // past the end: row kq, zeroed
#pragma unroll
        for (int i = 0; i < kFcG; ++i) {
            // k-steps past the end reuse the last valid one with an all-zero weight fragment (exact, see net_fc.hpp)
            const int ks = ks0 + i < nsteps ? ks0 + i : nsteps - 1;
            const int k = 4 * ks + kq;
            const int tap = k >> cin_log2, ci = k & ((1 << cin_log2) - 1);
            const int off = (tap / 3 - 1) * n.W + (tap % 3 - 1);
            rn_avec a;
#pragma unroll
            for (int s = 0; s < kRnMT; ++s) {
                const bool ok = tap < 9 && ((L.tapmask[s] >> tap) & 1u);
                const int row = ok ? L.m[s] + off : L.m[s];
                const float v = U[row * kRnLD + ci];
                a[s] = ok ? v : 0.0f;
            }
            EAZ_RN_MFMA(acc, a, bc.v[i]);
        }
        bc = bn;
    }
}

// 3x3 convolution over 64 input channels: the tap of a k-step is wave-uniform (k = 4*ks + kq, 64 channels per tap -> tap = ks / 16),
// so the neighbour row of every lane is resolved ONCE per tap -- off-board taps point at the all-zero row -- and the 16 k-steps
// of the tap are nothing but 4 ds_read_b32 with immediate offsets + 16 MFMAs each.
__device__ __forceinline__ void rn_conv64(floatx4 (&acc)[kRnMT][4], const ResNetDev& n, const float* U, const float* Wc, const RnLane& L, int lane) {
    const int r = lane & 15, kq = lane >> 4, col = 4 * r;
    BGroup b0, b1;
    // weights through a buffer resource: the lane's part of the address (row kq, columns col .. col + 3) never changes and the k-step goes into
    // the scalar offset -- a global_load needs three 64-bit VALU instructions per refill, and every VALU instruction between the MFMAs of this
    // loop costs the matrix pipe ~18 clocks
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wc), 0, 9 * kRnC * kRnC * 4, 0x00020000);
    const int wvoff = (kq * kRnC + col) * 4;
    auto load_b = [&](BGroup& b, int ks0) {
#pragma unroll
        for (int i = 0; i < kFcG; ++i) b.v[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff, (ks0 + i) * (4 * kRnC * 4), 0));
    };
    auto compute = [&](const BGroup& b, const float* const (&src)[kRnMT], int half) {
#pragma unroll
        for (int i = 0; i < kFcG; ++i) {
            const int ci0 = 4 * (half * 8 + i);
            rn_avec a;
#pragma unroll
            for (int s = 0; s < kRnMT; ++s) a[s] = src[s][ci0];
            EAZ_RN_MFMA(acc, a, b.v[i]);
        }
    };
    load_b(b0, 0);
#pragma clang loop unroll(disable)
    for (int tap = 0; tap < 9; ++tap) {
        const int off = (tap / 3 - 1) * n.W + (tap % 3 - 1);
        const float* src[kRnMT];
#pragma unroll
        for (int s = 0; s < kRnMT; ++s) {
            const bool ok = (L.tapmask[s] >> tap) & 1u;
            // U is the second tile: the all-zero row sits kRnRows rows behind U's first row
            src[s] = U + (ok ? L.m[s] + off : kRnZeroRow - kRnRows) * kRnLD + kq;
        }
        // weights of the next 8 k-steps are requested before the MFMAs of the current 8 and consumed after them (two register sets,
        // scheduling barriers pin the order; the very last request re-reads the last group: unconditional loads keep the
        // compiler's outstanding-load count exact, so it waits with vmcnt(8..1) instead of vmcnt(0))
        load_b(b1, tap * 16 + 8);
        __builtin_amdgcn_sched_barrier(kSchedPin);
        compute(b0, src, 0);
        __builtin_amdgcn_sched_barrier(kSchedPin);
        load_b(b0, tap < 8 ? tap * 16 + 16 : tap * 16 + 8);
        __builtin_amdgcn_sched_barrier(kSchedPin);
        compute(b1, src, 1);
        __builtin_amdgcn_sched_barrier(kSchedPin);
    }
}

__device__ __forceinline__ void rn_zero(floatx4 (&acc)[kRnMT][4]) {
#pragma unroll
    for (int s = 0; s < kRnMT; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
}

// accumulator element (s, t, reg) of lane (j = lane & 15, kq = lane >> 4) is row 16*MT*wave + MT*(4*kq + reg) + s, channel 4*j + t
#define EAZ_RN_FOREACH(BODY)                                                      \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                               \
        const int ch = 4 * (lane & 15) + t;                                       \
        _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                       \
            _Pragma("unroll") for (int s = 0; s < kRnMT; ++s) {                   \
                const int row = 16 * kRnMT * wave + kRnMT * (4 * (lane >> 4) + reg) + s; \
                BODY                                                              \
            }                                                                     \
    }

__global__ void __launch_bounds__(256, 2) resnet_forward_kernel(ResNetDev n, const uint8_t* __restrict__ obs, int B, NetOut out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* X = reinterpret_cast<float*>(lds);
    float* U = X + kRnRows * kRnLD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int s0 = blockIdx.x * kRnSamples;
    const int nsamp = B - s0 < kRnSamples ? B - s0 : kRnSamples;
    const int rows_valid = nsamp * n.P;
    const float* p = n.p;
    const RnLane L = rn_lane(n, wave, lane, rows_valid);
    const int D = n.P * n.Cin;
    EAZ_STAMP(0);
    if (tid < kRnLD) X[kRnZeroRow * kRnLD + tid] = 0.0f;

    // ---- input: observation bytes -> U[row][ci] as f32 (resnet.py:69) ----
    for (int i = tid; i < kRnRows * n.Cin; i += 256) {
        const int row = i / n.Cin, ci = i - row * n.Cin;
        float v = 0.0f;
        if (row < rows_valid) v = static_cast<float>(obs[static_cast<long>(s0) * D + static_cast<long>(row) * n.Cin + ci]);
        U[row * kRnLD + ci] = v;
    }
    __syncthreads();

    floatx4 acc[kRnMT][4];
    // ---- input conv (resnet.py:70): X = conv0(U) + b ----
    rn_zero(acc);
    rn_conv(acc, n, U, p + n.conv_w[0], n.conv_k[0], n.cin_log2, L, lane);
    {
        const float* b = p + n.conv_b[0];
        EAZ_RN_FOREACH(X[row * kRnLD + ch] = acc[s][t][reg] + b[ch];)
    }
    __syncthreads();

    EAZ_STAMP(1);
    // ---- residual tower: BlockV2 (resnet.py:31-39) ----
    for (int blk = 0; blk <= n.nb; ++blk) {
        // U = relu(bn(X)) with bn = bn1 of this block, or the final batch norm (resnet.py:79-81) when blk == nb
        {
            const int bi = 2 * blk;
            const float *mean = p + n.bn_mean[bi], *inv = p + n.bn_inv[bi], *off = p + n.bn_off[bi];
            for (int i = tid; i < kRnRows * kRnC; i += 256) {
                const int row = i >> 6, ch = i & 63;
                const float v = (X[row * kRnLD + ch] - mean[ch]) * inv[ch] + off[ch];
                U[row * kRnLD + ch] = v > 0.0f ? v : 0.0f;
            }
        }
        __syncthreads();
        if (blk == n.nb) break;
        // conv1, then U = relu(bn2(conv1 + b)) written in place once every wave has finished reading U
        rn_zero(acc);
        rn_conv64(acc, n, U, p + n.conv_w[1 + 2 * blk], L, lane);
        __syncthreads();
        {
            const float* b = p + n.conv_b[1 + 2 * blk];
            const float *mean = p + n.bn_mean[2 * blk + 1], *inv = p + n.bn_inv[2 * blk + 1], *off = p + n.bn_off[2 * blk + 1];
            EAZ_RN_FOREACH({
                const float v = ((acc[s][t][reg] + b[ch]) - mean[ch]) * inv[ch] + off[ch];
                U[row * kRnLD + ch] = v > 0.0f ? v : 0.0f;
            })
        }
        __syncthreads();
        // conv2 + skip: X = (conv2 + b) + X   (only this wave touches these rows of X)
        rn_zero(acc);
        rn_conv64(acc, n, U, p + n.conv_w[2 + 2 * blk], L, lane);
        {
            const float* b = p + n.conv_b[2 + 2 * blk];
            EAZ_RN_FOREACH(X[row * kRnLD + ch] = (acc[s][t][reg] + b[ch]) + X[row * kRnLD + ch];)
        }
        __syncthreads();
    }

    EAZ_STAMP(2);
    // ---- heads: 1x1 conv -> bn -> relu per cell, results into the (now free) X region ----
    // Hd layout: [sample][head][up to 84 floats] with head 0/1 = (cell, 2 planes) interleaved, head 2/3 = 42 cells
    float* Hd = X;
    constexpr int kHdStride = 84 * 2 + 42 * 2;  // floats per sample
    if (tid < rows_valid) {
        const int smp = tid / n.P, cell = tid - smp * n.P;
        const float* u = U + tid * kRnLD;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int nco = h < 2 ? 2 : 1;
            const float *w = p + n.hc_w[h], *b = p + n.hc_b[h];
            const float *mean = p + n.hbn_mean[h], *inv = p + n.hbn_inv[h], *off = p + n.hbn_off[h];
            for (int co = 0; co < nco; ++co) {
                float a = 0.0f;
#pragma unroll 8
                for (int ci = 0; ci < kRnC; ++ci) a = __builtin_fmaf(u[ci], w[ci * nco + co], a);
                float v = ((a + b[co]) - mean[co]) * inv[co] + off[co];
                v = v > 0.0f ? v : 0.0f;
                const int base = smp * kHdStride + (h < 2 ? h * 84 : 168 + (h - 2) * 42);
                Hd[base + cell * nco + co] = v;  // hk.Flatten keeps (h, w, c) order
            }
        }
    }
    __syncthreads();
    EAZ_STAMP(3);
    // linear layers: policy / exploration (84 -> A) and the first fc of value / ube (42 -> 64, relu) into U
    float* Fc = U;  // [sample][2][64]
    for (int i = tid; i < nsamp * (2 * n.A + 2 * kRnC); i += 256) {
        const int smp = i / (2 * n.A + 2 * kRnC), j = i - smp * (2 * n.A + 2 * kRnC);
        if (j < 2 * n.A) {
            const int h = j / n.A, a_ = j - h * n.A;
            const float *w = p + n.lin_w[h], *b = p + n.lin_b[h];
            const float* x = Hd + smp * kHdStride + h * 84;
            float z = 0.0f;
            for (int k = 0; k < 2 * n.P; ++k) z = __builtin_fmaf(x[k], w[k * n.A + a_], z);
            z = z + b[a_];
            float* dst = h == 0 ? out.exploit : out.explore;  // resnet.py:84-98
            if (dst) dst[static_cast<long>(s0 + smp) * n.A + a_] = z;
        } else {
            const int jj = j - 2 * n.A, h = jj >> 6, c = jj & 63;  // h: 0 value, 1 ube
            const float *w = p + n.lin_w[2 + 2 * h], *b = p + n.lin_b[2 + 2 * h];
            const float* x = Hd + smp * kHdStride + 168 + h * 42;
            float z = 0.0f;
            for (int k = 0; k < n.P; ++k) z = __builtin_fmaf(x[k], w[k * kRnC + c], z);
            z = z + b[c];
            Fc[(smp * 2 + h) * kRnC + c] = z > 0.0f ? z : 0.0f;  // resnet.py:105-106 / :117-118
        }
    }
    __syncthreads();
    EAZ_STAMP(4);
    if (tid < nsamp * 2) {
        const int smp = tid >> 1, h = tid & 1;
        const float *w = p + n.lin_w[3 + 2 * h], *b = p + n.lin_b[3 + 2 * h];
        const float* x = Fc + (smp * 2 + h) * kRnC;
        float z = 0.0f;
        for (int k = 0; k < kRnC; ++k) z = __builtin_fmaf(x[k], w[k], z);
        z = z + b[0];
        if (h == 0) {
            if (out.value) out.value[s0 + smp] = det_tanh(z);  // resnet.py:108
        } else {
            // resnet.py:120; the novelty max (:127-130) is applied by ube_combine() where the hash lookup is joined (max_u = 1)
            if (out.ube) out.ube[s0 + smp] = 0.5f * (det_tanh(z) + 1.0f);
        }
    }
    EAZ_STAMP(5);
}

}  // namespace eaz
