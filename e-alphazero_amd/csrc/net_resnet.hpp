// net_resnet.hpp -- EpistemicResidualAZNet forward (src/network/resnet.py:42-135; BlockV2 :26-39) for gfx950.
//
// Numerics contract (DESIGN.md "ResNet forward"), identical to the CPU oracle:
//   hk.Conv2D 3x3 SAME : acc = +0; for tap (kh, kw) row-major, ci ascending: acc = fmaf(x[nb][ci], w[tap][ci][co], acc),
//                        out-of-board taps contribute an exact 0; y = acc + b[co]          -> one f32 MFMA chain, K = 9*Cin
//   hk.BatchNorm (eval): y = (x - mean) * inv + offset, inv = scale / sqrt(var + 1e-5) precomputed on the host
//   hk.Linear / 1x1 conv: k-ascending fmaf chain, then + b
//
// Kernel shape (round 4): one 512-thread workgroup carries EIGHT samples (8 x 42 = 336 board cells = rows of the implicit GEMM = exactly
// 21 tiles of 16: no padding rows) through the WHOLE tower without touching HBM; one workgroup per CU, B = 8192 is 1024 workgroups =
// exactly four rounds on the 256 CUs.  Wave w owns output channels 16 (w & 3) .. + 15 for the rows of half w >> 2 (11 and 10 tiles): MT x 1
// tiles of v_mfma_f32_16x16x4_f32 per k-step on one B operand -- the two waves of a SIMD share 21 tiles per k-step -- so a wave streams a
// quarter of the weights (147 KB per 3x3 conv, shared by every workgroup, from L2 one group of 8 k-steps ahead: two 16-byte buffer loads,
// the host packs four consecutive k-steps of a lane side by side).  The residual stream X never leaves the accumulator layout (44
// registers per lane); only the activated conv input U lives in LDS ([336][65] f32 + one zero row, 88 KB) beside two 27 KB tables of the
// neighbour rows per tap (151 KB in all).  The A operand is gathered from U through the table's per-tap row addresses (MT ds_read_b32 per
// k-step, paired by the compiler), so no im2col buffer exists.  The two halves run in step (same barriers), so the input stage, the
// epilogues and the heads run at their own speed instead of beside a foreign MFMA stream.
// (Before: four samples per 256-thread workgroup, two per CU, 176 rows with 8 of padding: 1.875 ms at B = 8192; three samples, 2 x 4 tiles
// per wave, X and U in LDS: 1.996 ms; six samples / 4 x 4 tiles, one workgroup per CU: 2.87 ms.)
#pragma once
#include "net_fc.hpp"

namespace eaz {

constexpr int kRnSamples = 8;   // samples per workgroup (512 threads: wave w owns channels 16 (w & 3) .. of the rows of half w >> 2)
constexpr int kRnThreads = 512;
constexpr int kRnMT0 = 11, kRnMT1 = 10;  // 16-row MFMA tiles of the two row halves: 21 tiles = 336 rows = 8 x 42 cells, no padding rows
constexpr int kRnHalfRow = 16 * kRnMT0;  // first row of the second half
constexpr int kRnRows = 16 * (kRnMT0 + kRnMT1);  // GEMM rows per workgroup
constexpr int kRnLD = 65;       // LDS leading dimension (floats)
constexpr int kRnC = 64;        // channels the kernel is specialised for (resnet.py:48)
constexpr int kRnMaxConv = 33;  // conv0 + 2 * 16 blocks
constexpr int kRnZeroRow = kRnRows;  // one all-zero row behind the U tile: the source of every off-board tap
constexpr int kRnHdFloats = kRnSamples * (84 * 2 + 42 * 2);  // the heads' per-cell outputs, behind the zero row (before the tower: the rows' tap masks)
static_assert(kRnHdFloats >= kRnRows, "tap masks fit the heads' region");
constexpr int kRnHdOff = (((kRnRows + 1) * kRnLD * 4 + 15) / 16) * 16;  // byte offsets into the workgroup's LDS
constexpr int kRnTabOff = kRnHdOff + kRnHdFloats * 4;
// neighbour-row tables [2 halves][9 taps][64 lanes][12]: LDS address of (row of the lane's tile row s under the tap, or the zero row) + the
// lane's k quarter, built once per workgroup -- resolving the rows of a lane per tap in registers cost ~55 VALU instructions per tap, each
// beside the MFMA stream
constexpr int kRnTabBytes = 9 * 64 * 12 * 4;  // one half
constexpr int kRnLdsBytes = kRnTabOff + 2 * kRnTabBytes;  // 151 KB: one workgroup per CU
static_assert(kRnTabOff % 16 == 0 && kRnLdsBytes <= 160 * 1024, "ResNet LDS layout");

struct ResNetDev {
    int H, W, Cin, cin_log2, nb, A, P;
    const float* p;  // packed parameters (device)
    int conv_w[kRnMaxConv], conv_b[kRnMaxConv], conv_k[kRnMaxConv];  // conv 0 = input conv, then (conv1, conv2) per block; K padded to 4
    int bn_mean[kRnMaxConv], bn_inv[kRnMaxConv], bn_off[kRnMaxConv];  // bn1 / bn2 per block at 2k, 2k+1; final bn at 2*nb
    int hc_w[4], hc_b[4], hbn_mean[4], hbn_inv[4], hbn_off[4];        // heads: policy, exploration, value, ube
    int lin_w[6], lin_b[6];                                            // policy, exploration, value fc1, value fc2, ube fc1, ube fc2
};

// Device layout of a 64 -> 64 3x3 convolution's weights (host: resnet_pack_tower_conv in eaz_api.hip): the B operand of k-step ks for lane
// (kq = lane >> 4, channel c) is W[4 ks + kq][c]; four consecutive k-steps of one lane sit side by side, so one 16-byte load serves four:
//   Wp[((ks4 * 4 + kq) * 64 + c) * 4 + q] = W[(4 * (4 * ks4 + q) + kq) * 64 + c]
constexpr int kRnPackBytes = 4 * kRnC * 4 * 4;  // one group of four k-steps

// rows of this lane as an A operand: m_s = MT*i + s  (i = lane & 15, s < MT); a 9-bit mask of the taps of the row's cell (h, w) that stay
// on the board
template <int MT>
struct RnLane {
    int m[MT];
    unsigned tapmask[MT];
};

// The tap mask of ONE row (thread t < kRnRows computes row t's; the division by the board's width is not cheap, and a lane needs eleven)
__device__ __forceinline__ unsigned rn_row_mask(const ResNetDev& n, int m, int rows_valid) {
    unsigned mask = 0;
    if (m < rows_valid) {
        const int cell = m % n.P, h = cell / n.W, w = cell - h * n.W;
        for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
            if (hh >= 0 && hh < n.H && ww >= 0 && ww < n.W) mask |= 1u << t;
        }
    }
    return mask;
}

template <int MT>
__device__ __forceinline__ RnLane<MT> rn_lane(const uint32_t* row_mask /* LDS, [kRnRows] */, int rbase, int lane) {
    RnLane<MT> L;
#pragma unroll
    for (int s = 0; s < MT; ++s) {
        L.m[s] = rbase + MT * (lane & 15) + s;
        L.tapmask[s] = row_mask[L.m[s]];
    }
    return L;
}

#define EAZ_RN_MFMA(acc, a, b) \
    _Pragma("unroll") for (int s_ = 0; s_ < MT; ++s_) acc[s_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], b, acc[s_], 0, 0, 0)

// the input convolution (any Cin, K = 9 * Cin padded to 4) as an implicit GEMM: acc[s] += gather(U)[rows][k] * Wc[k][16 wave + j]; weights in
// the plain [K][64] layout.  k-steps past the end use an all-zero weight fragment (exact, see net_fc.hpp)
template <int MT>
__device__ __forceinline__ void rn_conv(floatx4 (&acc)[MT], const ResNetDev& n, const float* U, const float* Wc, int K, int cin_log2,
                                        const RnLane<MT>& L, int wave, int lane) {
    const int kq = lane >> 4, col = 16 * (wave & 3) + (lane & 15);
    const int nsteps = K >> 2;
    for (int ks = 0; ks < nsteps; ++ks) {
        const int k = 4 * ks + kq;
        const float b = Wc[static_cast<long>(k) * kRnC + col];
        const int tap = k >> cin_log2, ci = k & ((1 << cin_log2) - 1);
        const int off = (tap / 3 - 1) * n.W + (tap % 3 - 1);
        float a[MT];
#pragma unroll
        for (int s = 0; s < MT; ++s) {
            const bool ok = tap < 9 && ((L.tapmask[s] >> tap) & 1u);
            const int row = ok ? L.m[s] + off : L.m[s];
            const float v = U[row * kRnLD + ci];
            a[s] = ok ? v : 0.0f;
        }
        EAZ_RN_MFMA(acc, a, b);
    }
}

// 3x3 convolution over 64 input channels: the tap of a k-step is wave-uniform (k = 4*ks + kq, 64 channels per tap -> tap = ks / 16),
// so the neighbour rows of a lane change ONCE per tap -- three 16-byte reads of the table, requested a tap ahead; off-board taps point at
// the all-zero row -- and the 16 k-steps of the tap are nothing but 11 ds_read_b32 with immediate offsets + 11 MFMAs each.
template <int MT>
__device__ __forceinline__ void rn_conv64(floatx4 (&acc)[MT], const uint8_t* lds, const float* Wc, int wave, int lane) {
    const int kq = lane >> 4;
    floatx4 b0[2], b1[2];
    // weights through a buffer resource: the lane's part of the address never changes and the k-step group goes into the scalar offset -- a
    // global_load needs three 64-bit VALU instructions per refill, and every VALU instruction between the MFMAs of this loop costs the
    // matrix pipe ~18 clocks
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wc), 0, 9 * kRnC * kRnC * 4, 0x00020000);
    const int wvoff = (kq * kRnC + 16 * (wave & 3) + (lane & 15)) * 16;
    auto load_b = [&](floatx4 (&b)[2], int ks0) {  // the 8 k-steps ks0 .. ks0 + 7 (ks0 a multiple of 8)
#pragma unroll
        for (int g = 0; g < 2; ++g) b[g] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff, ((ks0 >> 2) + g) * kRnPackBytes, 0));
    };
    const uint4* tab = reinterpret_cast<const uint4*>(lds + kRnTabOff + (wave >> 2) * kRnTabBytes) + lane * 3;
    auto load_t = [&](uint4 (&t)[3], int tap) {
#pragma unroll
        for (int g = 0; g < 3; ++g) t[g] = tab[tap * (64 * 3) + g];
    };
    auto compute = [&](const floatx4 (&b)[2], const uint4 (&t)[3], int half) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ci0 = 4 * (half * 8 + i);
            float a[MT];
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                const uint32_t o = s & 3 ? (s & 3) == 1 ? t[s >> 2].y : (s & 3) == 2 ? t[s >> 2].z : t[s >> 2].w : t[s >> 2].x;
                a[s] = ((const __attribute__((address_space(3))) float*)o)[ci0];  // `o` is the LDS address itself: no add per row and tap
            }
            EAZ_RN_MFMA(acc, a, b[i >> 2][i & 3]);
        }
    };
    // weights of the next 8 k-steps are requested before the MFMAs of the current 8 and consumed after them (two register sets,
    // scheduling barriers pin the order; the very last request re-reads the last group: unconditional loads keep the
    // compiler's outstanding-load count exact, so it waits with a counted vmcnt instead of vmcnt(0)); the row addresses likewise: two sets,
    // the taps taken in pairs so that no set is ever copied
    auto one_tap = [&](int tap, const uint4 (&tc)[3], uint4 (&tn)[3]) {
        load_t(tn, tap < 8 ? tap + 1 : tap);
        __builtin_amdgcn_sched_barrier(0);  // issued here, a tap ahead: left to the scheduler the three reads sink to the end of the tap, in front of their first use
        load_b(b1, tap * 16 + 8);
        __builtin_amdgcn_sched_barrier(kSchedPin);
        compute(b0, tc, 0);
        __builtin_amdgcn_sched_barrier(kSchedPin);
        load_b(b0, tap < 8 ? tap * 16 + 16 : tap * 16 + 8);
        __builtin_amdgcn_sched_barrier(kSchedPin);
        compute(b1, tc, 1);
        __builtin_amdgcn_sched_barrier(kSchedPin);
    };
    uint4 t0[3], t1[3];
    load_b(b0, 0);
    load_t(t0, 0);
#pragma clang loop unroll(disable)
    for (int tap = 0; tap < 8; tap += 2) {
        one_tap(tap, t0, t1);
        one_tap(tap + 1, t1, t0);
    }
    one_tap(8, t0, t1);
}

template <int MT>
__device__ __forceinline__ void rn_zero(floatx4 (&acc)[MT]) {
#pragma unroll
    for (int s = 0; s < MT; ++s) acc[s] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
}

// accumulator element (s, reg) of lane (j = lane & 15, kq = lane >> 4) is row rbase + MT*(4*kq + reg) + s, channel `ch` = 16*(wave & 3) + j
#define EAZ_RN_FOREACH(BODY)                                              \
    _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                   \
        _Pragma("unroll") for (int s = 0; s < MT; ++s) {                  \
            const int row = rbase + MT * (4 * (lane >> 4) + reg) + s;     \
            BODY                                                          \
        }

// input conv + residual tower of one wave: the rows of its half (MT tiles from row `rbase`) for its 16 channels; leaves U = relu(bn_final(X))
template <int MT>
__device__ __forceinline__ void rn_tower(const ResNetDev& n, uint8_t* lds, const uint32_t* row_mask, const int rbase, const int tid, const int wave, const int lane) {
    float* U = reinterpret_cast<float*>(lds);
    const float* p = n.p;
    const int ch = 16 * (wave & 3) + (lane & 15);  // the channel of this lane's accumulator elements
    const RnLane<MT> L = rn_lane<MT>(row_mask, rbase, lane);
    {
        // the neighbour-row table of rn_conv64 (the same for the four waves of a half: wave w fills taps w & 3, + 4, + 8)
        uint32_t* T = reinterpret_cast<uint32_t*>(lds + kRnTabOff + (wave >> 2) * kRnTabBytes);
        for (int t = wave & 3; t < 9; t += 4) {
            const int off = (t / 3 - 1) * n.W + (t % 3 - 1);
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                const bool ok = (L.tapmask[s] >> t) & 1u;
                // the low half of an LDS location's flat address is its LDS address
                T[(t * 64 + lane) * 12 + s] = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(U + (ok ? L.m[s] + off : kRnZeroRow) * kRnLD + (lane >> 4)));
            }
#pragma unroll
            for (int s = MT; s < 12; ++s) T[(t * 64 + lane) * 12 + s] = 0u;
        }
    }
    floatx4 acc[MT], xr[MT];  // xr: the residual stream X, in the accumulator layout for the whole tower
    // ---- input conv (resnet.py:70): X = conv0(U) + b ----
    rn_zero<MT>(acc);
    rn_conv<MT>(acc, n, U, p + n.conv_w[0], n.conv_k[0], n.cin_log2, L, wave, lane);
    {
        const float b = p[n.conv_b[0] + ch];
#pragma unroll
        for (int s = 0; s < MT; ++s)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) xr[s][reg] = acc[s][reg] + b;
    }
    __syncthreads();  // every wave has read the observation out of U

    EAZ_STAMP(1);
    // ---- residual tower: BlockV2 (resnet.py:31-39) ----
    for (int blk = 0; blk <= n.nb; ++blk) {
        // U = relu(bn(X)) with bn = bn1 of this block, or the final batch norm (resnet.py:79-81) when blk == nb
        {
            const int bi = 2 * blk;
            const float mean = p[n.bn_mean[bi] + ch], inv = p[n.bn_inv[bi] + ch], off = p[n.bn_off[bi] + ch];
            EAZ_RN_FOREACH({
                const float v = (xr[s][reg] - mean) * inv + off;
                U[row * kRnLD + ch] = v > 0.0f ? v : 0.0f;
            })
        }
        __syncthreads();
        if (blk == n.nb) break;
        // conv1, then U = relu(bn2(conv1 + b)) written in place once every wave has finished reading U
        rn_zero<MT>(acc);
        rn_conv64<MT>(acc, lds, p + n.conv_w[1 + 2 * blk], wave, lane);
        __syncthreads();
        {
            const float b = p[n.conv_b[1 + 2 * blk] + ch];
            const float mean = p[n.bn_mean[2 * blk + 1] + ch], inv = p[n.bn_inv[2 * blk + 1] + ch], off = p[n.bn_off[2 * blk + 1] + ch];
            EAZ_RN_FOREACH({
                const float v = ((acc[s][reg] + b) - mean) * inv + off;
                U[row * kRnLD + ch] = v > 0.0f ? v : 0.0f;
            })
        }
        __syncthreads();
        // conv2 + skip: X = (conv2 + b) + X, in registers
        rn_zero<MT>(acc);
        rn_conv64<MT>(acc, lds, p + n.conv_w[2 + 2 * blk], wave, lane);
        {
            const float b = p[n.conv_b[2 + 2 * blk] + ch];
#pragma unroll
            for (int s = 0; s < MT; ++s)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) xr[s][reg] = (acc[s][reg] + b) + xr[s][reg];
        }
        __syncthreads();  // every wave has finished reading U before the next block's bn1 overwrites it
    }
}

__global__ void __launch_bounds__(kRnThreads, 1) resnet_forward_kernel(ResNetDev n, const uint8_t* __restrict__ obs, int B, NetOut out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* U = reinterpret_cast<float*>(lds);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int s0 = blockIdx.x * kRnSamples;
    const int nsamp = B - s0 < kRnSamples ? B - s0 : kRnSamples;
    const int rows_valid = nsamp * n.P;
    const float* p = n.p;
    // the rows' tap masks, one row per thread, through the (still unused) heads' region
    uint32_t* row_mask = reinterpret_cast<uint32_t*>(lds + kRnHdOff);
    if (tid < kRnRows) row_mask[tid] = rn_row_mask(n, tid, rows_valid);
    const int D = n.P * n.Cin;
    EAZ_STAMP(0);
    if (tid < kRnLD) U[kRnZeroRow * kRnLD + tid] = 0.0f;
    // ---- input: observation bytes -> U[row][ci] as f32 (resnet.py:69) ----
    for (int i = tid; i < kRnRows * n.Cin; i += kRnThreads) {
        const int row = i / n.Cin, ci = i - row * n.Cin;
        float v = 0.0f;
        if (row < rows_valid) v = static_cast<float>(obs[static_cast<long>(s0) * D + static_cast<long>(row) * n.Cin + ci]);
        U[row * kRnLD + ci] = v;
    }
    __syncthreads();
    // the two row halves run the same sequence of barriers (the second has one tile less per k-step)
    if (wave < 4) rn_tower<kRnMT0>(n, lds, row_mask, 0, tid, wave, lane);
    else rn_tower<kRnMT1>(n, lds, row_mask, kRnHalfRow, tid, wave, lane);

    EAZ_STAMP(2);
    // ---- heads: 1x1 conv -> bn -> relu per cell, results into their own LDS region ----
    // Hd layout: [sample][head][up to 84 floats] with head 0/1 = (cell, 2 planes) interleaved, head 2/3 = 42 cells
    float* Hd = reinterpret_cast<float*>(lds + kRnHdOff);  // behind the zero row
    constexpr int kHdStride = 84 * 2 + 42 * 2;  // floats per sample
    // On MFMA, like the tower (the k-ascending chain of a v_mfma_f32_16x16x4_f32 IS the oracle's fmaf chain): as VALU chains beside the co-resident
    // workgroup's MFMA stream the heads took 21 % of a workgroup's timeline, and two workgroups that start together reach them together.
    const int hj = lane & 15, hkq = lane >> 4;
    {
        // columns of the 1x1 convolutions side by side: policy 0-1, exploration 2-3, value 4, ube 5 (B operand: W[k][column], zero beyond)
        const int hh = hj < 2 ? 0 : hj < 4 ? 1 : hj - 2, hco = hj < 4 ? hj & 1 : 0, hnco = hj < 4 ? 2 : 1;
        const bool hcol = hj < 6;
        const int h_ = hcol ? hh : 0;
        float wb[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float v = p[n.hc_w[h_] + (4 * ks + hkq) * hnco + hco];
            wb[ks] = hcol ? v : 0.0f;
        }
        const float hb = p[n.hc_b[h_] + hco], hmean = p[n.hbn_mean[h_] + hco], hinv = p[n.hbn_inv[h_] + hco], hoff = p[n.hbn_off[h_] + hco];
        for (int T = wave; T < kRnMT0 + kRnMT1; T += 8) {
            const float* u = U + (16 * T + hj) * kRnLD + hkq;
            floatx4 d = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(u[4 * ks], wb[ks], d, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * T + 4 * hkq + reg;
                if (hcol && row < rows_valid) {
                    int smp = 0;
#pragma unroll
                    for (int q = 1; q < kRnSamples; ++q) smp += row >= q * n.P;
                    const int cell = row - smp * n.P;
                    float v = ((d[reg] + hb) - hmean) * hinv + hoff;
                    v = v > 0.0f ? v : 0.0f;
                    const int base = smp * kHdStride + (hh < 2 ? hh * 84 : 168 + (hh - 2) * 42);
                    Hd[base + cell * hnco + hco] = v;  // hk.Flatten keeps (h, w, c) order
                }
            }
        }
    }
    __syncthreads();
    EAZ_STAMP(3);
    // linear layers: policy / exploration (84 -> A; waves 0 / 1) and the first fc of value / ube (42 -> 64, relu; waves 2 / 3) into U.
    // Rows of the MFMA tile = the workgroup's samples (4 of 16 used), K padded to a multiple of 4 with zero operands (fmaf(0, 0, acc) = acc).
    float* Fc = U;  // [sample][2][64]
    if (wave < 2 && n.A <= 16) {
        const int h = wave, K = 2 * n.P;
        const float *w = p + n.lin_w[h], *b = p + n.lin_b[h];
        const float* x = Hd + (hj < nsamp ? hj : 0) * kHdStride + h * 84;
        floatx4 d = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        float av[21], bv[21];  // K <= 84 (the host refuses boards of more than 42 cells): every operand requested before the chain starts
#pragma unroll
        for (int i = 0; i < 21; ++i) {
            const int k = 4 * i + hkq;
            const bool ok = k < K;
            const float xa = x[ok ? k : 0], wv = w[(ok ? k : 0) * n.A + (hj < n.A ? hj : 0)];
            av[i] = ok && hj < nsamp ? xa : 0.0f;
            bv[i] = ok && hj < n.A ? wv : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 21; ++i) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], d, 0, 0, 0);  // all-zero k-steps past K leave d as it is
        float* dst = h == 0 ? out.exploit : out.explore;  // resnet.py:84-98
        if (dst && hkq < kRnSamples / 4 && hj < n.A) {
            const float bb = b[hj];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (4 * hkq + reg < nsamp) dst[static_cast<long>(s0 + 4 * hkq + reg) * n.A + hj] = d[reg] + bb;
        }
    } else if (wave == 2 || wave == 3) {
        const int h = wave - 2, K = n.P;  // h: 0 value, 1 ube
        const float *w = p + n.lin_w[2 + 2 * h], *b = p + n.lin_b[2 + 2 * h];
        const float* x = Hd + (hj < nsamp ? hj : 0) * kHdStride + 168 + h * 42;
        floatx4 d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) d[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        float av[11], bv[11][4];  // K <= 42
#pragma unroll
        for (int ks = 0; ks < 11; ++ks) {
            const int k = 4 * ks + hkq;
            const bool ok = k < K;
            const float xa = x[ok ? k : 0];
            av[ks] = ok && hj < nsamp ? xa : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float wv = w[(ok ? k : 0) * kRnC + 16 * t + hj];
                bv[ks][t] = ok ? wv : 0.0f;
            }
        }
#pragma unroll
        for (int ks = 0; ks < 11; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks][t], d[t], 0, 0, 0);
        if (hkq < kRnSamples / 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = 16 * t + hj;
                const float bb = b[c];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float z = d[t][reg] + bb;
                    if (4 * hkq + reg < nsamp) Fc[((4 * hkq + reg) * 2 + h) * kRnC + c] = z > 0.0f ? z : 0.0f;  // resnet.py:105-106 / :117-118
                }
            }
        }
    }
    if (n.A > 16) {
        // more actions than one MFMA tile has columns: the policy / exploration layers as plain chains
        for (int i = tid; i < nsamp * 2 * n.A; i += kRnThreads) {
            const int smp = i / (2 * n.A), j = i - smp * (2 * n.A);
            const int h = j / n.A, a_ = j - h * n.A;
            const float *w = p + n.lin_w[h], *b = p + n.lin_b[h];
            const float* x = Hd + smp * kHdStride + h * 84;
            float z = 0.0f;
            for (int k = 0; k < 2 * n.P; ++k) z = __builtin_fmaf(x[k], w[k * n.A + a_], z);
            z = z + b[a_];
            float* dst = h == 0 ? out.exploit : out.explore;
            if (dst) dst[static_cast<long>(s0 + smp) * n.A + a_] = z;
        }
    }
    __syncthreads();
    EAZ_STAMP(4);
    if (tid < nsamp * 2) {
        const int smp = tid >> 1, h = tid & 1;
        const float *w = p + n.lin_w[3 + 2 * h], *b = p + n.lin_b[3 + 2 * h];
        const float* x = Fc + (smp * 2 + h) * kRnC;
        float z = 0.0f;
#pragma unroll 16
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
