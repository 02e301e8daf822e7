// net_fc_pipe.hpp -- fc16p_forward_kernel: the FC forward of one head for a 16-row tile of dense observations, weights streamed exactly like
// the persistent search kernel's layer 2 (search_persist.hpp ps_round).
//
// Reference: EpistemicFullyConnectedAZNet.__call__, src/network/fully_connected.py:41-101.  Numerics: identical to fc16_forward_kernel
// and the oracle (layers 1-2 one k-ascending chain per output, layer 3 four quarter chains, net_fc.hpp header).
//
// Why a second 16-row kernel: fc16_forward_kernel spreads (tile, head) pairs over 3-4 co-resident workgroups per CU and leaves the weight
// loads to the compiler's schedule; measured 0.44-0.49 of the f32-MFMA peak on dense rows (C3: 44 us per launch).  The persistent search
// kernel's loop -- one wave per SIMD, weights 16 k-steps ahead as buffer loads with the k-step in the scalar offset, every refill pinned
// behind the k-step that freed its registers -- runs at 0.92 of the MFMA issue rate.  This kernel is that loop for dense observations:
// the uint8 tile is converted once to an f32 K-major LDS tile, and the weight stream runs on from layer 1 into layer 2, so only the very
// first 16 k-steps of a workgroup wait for memory (and three workgroups share a CU).  Layer 1 runs whole
// rounds of 16 k-steps: the device copy of W1 carries zero rows up to fp_kpad(in_dim) (eaz_net_create_fc) and the tile is zero there
// too (exact: fma(0, 0, acc) = acc), so in_dim needs no alignment and no read leaves the allocation.
#pragma once
#include "search_persist.hpp"

namespace eaz {

constexpr int kFpMaxK = 1024;  // layer-1 inputs the f32 tile holds (70 KB of LDS at 1024)
__host__ __device__ inline int fp_kpad(int K) { return ((K + 63) / 64) * 64; }  // whole rounds of 16 k-steps
__host__ __device__ inline int fp_xfloats(int K) { const int kp = fp_kpad(K); return kp * 16 + (kp / 4) * 4; }
// LDS: the observation tile (later reused as the H2 tile: it is dead once layer 1 is done), H1, the layer-3 quarter sums
__host__ __device__ inline int fp_lds_bytes(int K, int A) {
    const int x = fp_xfloats(K) * 4;
    return (x > kF16HBytes ? x : kF16HBytes) + kF16HBytes + 4 * 16 * 16 * 4;
}
// dynamic LDS of subleq16_search_kernel (search_subleq.hpp): the forward tiles of its head groups + the shared observation tile
__host__ __device__ inline int sq16_lds_bytes(int in_dim, int A, int head_groups) { return head_groups * fp_lds_bytes(in_dim, A) + fp_xfloats(in_dim) * 4; }
__host__ inline bool fp_eligible(int in_dim, int A) { return in_dim <= kFpMaxK && A <= kFcW3Cols && kPsDepth == 16; }
// workgroup barriers inside fc16p_body: a caller whose other wavefronts do not run the body must issue exactly this many (search_subleq*.hpp);
// tests/test_isa_guard.py counts the s_barrier instructions of fc16p_forward_kernel against it
constexpr int kFpBarriers = 3;

// grid = (row tiles, heads): one workgroup = 16 rows x one head (40 KB of LDS at the C3 shape, 148 VGPRs: three workgroups per CU, whose
// barriers / staging / layer-3 chains overlap each other's MFMA streams)
// The forward of head `head` for the 16 rows from `row0` on, by 256 threads `tid` = 0..255 (three workgroup barriers; `lds` = fp_lds_bytes).
// Called once per workgroup by fc16p_forward_kernel and once per head and simulation by the persistent Subleq search kernel.
// fp_convert_tile: the uint8 observation rows row0 .. row0 + 15 -> the f32 K-major tile X (the MFMA A-operand layout of the hidden tiles),
// zero beyond in_dim / beyond the batch; `nthreads` threads `tid` share the work.  fc16p_body does it itself unless the caller hands it a
// tile (Xshared): the persistent Subleq search converts the 16 new rows once per simulation for all three heads.
__device__ __forceinline__ void fp_convert_tile(const uint8_t* __restrict__ obs, const int B, const int K1, const int row0, float* X, const int tid,
                                                const int nthreads) {
    const int Kp = fp_kpad(K1);
    if ((K1 & 15) == 0 && (reinterpret_cast<uintptr_t>(obs) & 15) == 0) {
        const int cpr = Kp >> 4, nch = kF16M * cpr;
        for (int c = tid; c < nch; c += nthreads) {
            const int row = c / cpr, k0 = (c - row * cpr) << 4;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row0 + row < B && k0 < K1) v = *reinterpret_cast<const uint4*>(obs + static_cast<long>(row0 + row) * K1 + k0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 16; ++i) X[h16_off(k0 + i, row)] = static_cast<float>((w[i >> 2] >> (8 * (i & 3))) & 0xFFu);
        }
    } else {
        for (int row = 0; row < kF16M; ++row) {
            const bool rok = row0 + row < B;
            const uint8_t* orow = obs + static_cast<long>(rok ? row0 + row : 0) * K1;
            for (int k = tid; k < Kp; k += nthreads) X[h16_off(k, row)] = (rok && k < K1) ? static_cast<float>(orow[k]) : 0.0f;
        }
    }
}

__device__ __forceinline__ void fc16p_body(const NetDev& net, const uint8_t* __restrict__ obs, const int B, const NetOut& out, const int head,
                                           const int row0, uint8_t* lds, const int tid, unsigned long long* kstep_counter,
                                           const float* Xshared = nullptr) {
    const int K1 = net.in_dim, Kp = fp_kpad(K1);
    float* Xown = reinterpret_cast<float*>(lds);
    const float* X = Xshared ? Xshared : Xown;
    float* H2 = Xown;  // aliases the workgroup's own observation tile (dead once layer 1 is done); a shared tile is left alone
    const int xf = fp_xfloats(K1) > kF16HFloats ? fp_xfloats(K1) : kF16HFloats;
    float* H1 = Xown + xf;
    float* part = H1 + kF16HFloats;      // [4 waves][16 rows][16 outputs]
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int n0 = wave * 64, col = n0 + 4 * r, voff = (kq * kFcH + col) * 4;
    const int No = head < 2 ? 1 : net.A;
    auto rsrc = [](const float* p, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rs1 = rsrc(net.w[head * 3], Kp * kFcH * 4), rs2 = rsrc(net.w[head * 3 + 1], kFcH * kFcH * 4);

    EAZ_STAMP(0);
    floatx4 Bq[kPsDepth];
    // biases of this lane's 4 columns; layer-3 weights of this wave's quarter of k as MFMA B operands (W3[n0 + 4q + kq][r], columns beyond
    // the head's outputs zero); the output bias of the (row, output) this thread emits
    float b1v[4], b2v[4], w3v[16];
    const float *pb1 = net.b[head * 3] + col, *pb2 = net.b[head * 3 + 1] + col, *pb3 = net.b[head * 3 + 2];
    const float* pw3 = net.w[head * 3 + 2] + (n0 + kq) * No + (r < No ? r : 0);  // pointers out of the kernel arguments once, not per load
#pragma unroll
    for (int c = 0; c < 4; ++c) { b1v[c] = pb1[c]; b2v[c] = pb2[c]; }
#pragma unroll
    for (int q = 0; q < 16; ++q) { const float w = pw3[4 * q * No]; w3v[q] = r < No ? w : 0.0f; }
    const int erow = tid & 15, eo = tid >> 4;
    const float b3v = pb3[eo < No ? eo : 0];
    // observation tile -> f32, K-major (the MFMA A-operand layout of the hidden tiles), zero beyond in_dim / beyond the batch
    if (Xshared) {
#pragma unroll
        for (int i = 0; i < kPsDepth; ++i) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, i * 4096, 0));
    } else if ((K1 & 15) == 0 && (reinterpret_cast<uintptr_t>(obs) & 15) == 0) {
        // 16 bytes per load: chunk c of the tile = row c / (Kp / 16), inputs 16 (c % (Kp / 16)) .. +15
        const int cpr = Kp >> 4, nch = kF16M * cpr;
        auto chunk_load = [&](int c) {
            const int row = c / cpr, k0 = (c - row * cpr) << 4;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (c < nch && row0 + row < B && k0 < K1) v = *reinterpret_cast<const uint4*>(obs + static_cast<long>(row0 + row) * K1 + k0);
            return v;
        };
        auto chunk_store = [&](int c, const uint4& v) {
            if (c >= nch) return;
            const int row = c / cpr, k0 = (c - row * cpr) << 4;
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 16; ++i) Xown[h16_off(k0 + i, row)] = static_cast<float>((w[i >> 2] >> (8 * (i & 3))) & 0xFFu);
        };
        const uint4 v0 = chunk_load(tid);
        // the weight stream's first 16 k-steps go out BEHIND the tile / constant loads: vmcnt is in order, and the tile must not wait for
        // 16 KB of weights per wave while every workgroup of the launch starts its stream at once
#pragma unroll
        for (int i = 0; i < kPsDepth; ++i) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, i * 4096, 0));
        chunk_store(tid, v0);
        for (int c = tid + 256; c < nch; c += 256) chunk_store(c, chunk_load(c));
    } else {
#pragma unroll
        for (int i = 0; i < kPsDepth; ++i) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, i * 4096, 0));
        for (int row = 0; row < kF16M; ++row) {
            const bool rok = row0 + row < B;
            const uint8_t* orow = obs + static_cast<long>(rok ? row0 + row : 0) * K1;
            for (int k = tid; k < Kp; k += 256) Xown[h16_off(k, row)] = (rok && k < K1) ? static_cast<float>(orow[k]) : 0.0f;
        }
    }
    __syncthreads();  // X
    EAZ_STAMP(1);
    floatx4 dummy = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    floatx4 acc[4];
    auto store_tile = [&](float* Ht, const float (&bv)[4]) {
        // relu(acc + bias) -> hidden tile: lane (r, kq) holds rows 4 kq .. +3 of column n0 + 4r + c
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            floatx4 o;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float v = acc[c][reg] + bv[c];
                o[reg] = v > 0.0f ? v : 0.0f;
            }
            *reinterpret_cast<floatx4*>(Ht + h16_off(col + c, 4 * kq)) = o;
        }
    };
    // ---------------- layer 1: [16 x K1] x [K1 x 256] ----------------
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    {
        const int rounds1 = Kp / 64;
        const float* Ap = X + kq * 16 + r;
        float av[kPsARing];
#pragma unroll
        for (int i = 0; i < kPsARing; ++i) av[i] = Ap[i * 68];
#pragma unroll 1
        for (int rd = 0; rd < rounds1; ++rd) {
            const bool same = rd + 1 < rounds1;  // the refills of the last round are the first k-steps of layer 2
            ps_round<false>(acc, Bq, av, Ap + (rd * kPsDepth + kPsARing) * 68, same ? rs1 : rs2, voff, same ? 4096 * kPsDepth * (rd + 1) : 0,
                            dummy, nullptr, nullptr);
        }
    }
    EAZ_STAMP(2);
    store_tile(H1, b1v);
    __syncthreads();  // H1 complete; every wave is done with the observation tile
    EAZ_STAMP(3);
    // ---------------- layer 2: [16 x 256] x [256 x 256] ----------------
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    {
        const float* Ap = H1 + kq * 16 + r;
        float av[kPsARing];
#pragma unroll
        for (int i = 0; i < kPsARing; ++i) av[i] = Ap[i * 68];
#pragma unroll 1
        for (int rd = 0; rd < 64 / kPsDepth; ++rd) {
            const bool same = rd + 1 < 64 / kPsDepth;  // the refills of the last round are never used: harmless re-reads of W2's first rows
            ps_round<false>(acc, Bq, av, Ap + (rd * kPsDepth + kPsARing) * 68, rs2, voff, same ? 4096 * kPsDepth * (rd + 1) : 0, dummy, nullptr, nullptr);
        }
    }
    EAZ_STAMP(4);
    store_tile(H2, b2v);
    // ---------------- layer 3: quarter chain over the 64 hidden columns this wave owns, then ((c0 + c1) + (c2 + c3)) + b ----------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // only columns written by this wave are read back
    {
        const float* hpA = H2 + h16_off(n0 + kq, r);
        floatx4 a3 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 16; ++q) a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(hpA[68 * q], w3v[q], a3, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) part[(wave * 16 + 4 * kq + reg) * 16 + r] = a3[reg];
    }
    __syncthreads();
    EAZ_STAMP(5);
    if (eo < No && row0 + erow < B) {
        const float c0 = part[erow * 16 + eo], c1 = part[(16 + erow) * 16 + eo], c2 = part[(32 + erow) * 16 + eo], c3 = part[(48 + erow) * 16 + eo];
        const float z = ((c0 + c1) + (c2 + c3)) + b3v;
        const long gi = row0 + erow;
        if (head == 0) { if (out.value) out.value[gi] = det_tanh(z); }                                   // fully_connected.py:55
        else if (head == 1) { if (out.ube) out.ube[gi] = (0.5f * (det_tanh(z) + 1.0f)) * net.max_u; }    // :64, :93 (:95-96 in ube_combine)
        else { float* dst = head == 2 ? out.exploit : out.explore; if (dst) dst[gi * net.A + eo] = z; }  // :68-81
    }
    EAZ_STAMP(6);
    if (kstep_counter && tid == 0) atomicAdd(kstep_counter, static_cast<unsigned long long>((K1 + 3) / 4));
}

__global__ void __launch_bounds__(256) fc16p_forward_kernel(NetDev net, const uint8_t* __restrict__ obs, int B, NetOut out, HeadList heads,
                                                            unsigned long long* kstep_counter) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    fc16p_body(net, obs, B, out, heads.id[blockIdx.y], blockIdx.x * kF16M, lds, threadIdx.x, kstep_counter);
}

}  // namespace eaz
