// mfma_issue_probe.hip -- does a second MFMA wavefront per SIMD close the issue gaps of the layer-2 loop of search_persist_kernel?
// The loop under test (search_persist.hpp ps_round): per k-step 4 x v_mfma_f32_16x16x4_f32 on weights prefetched 16 k-steps ahead, one
// buffer_load_dwordx4 refill, one ds_read_b32 of the next A operand; one wavefront per SIMD, each owning 64 of the 256 hidden columns.
// Variant 8: eight wavefronts (two per SIMD), each owning 32 columns: 2 MFMAs per k-step, one dwordx4 refill per TWO k-steps.
// Prints shader clocks per "head" (64 k-steps x 256 columns x 16 rows) for both; the MFMA floor is 64 x 4 x 32 = 8192.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/mfma_issue_probe tools/mfma_issue_probe.hip && tools/bin/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kDepth = 16;

template <int WAVES, int VAR>
__global__ void __launch_bounds__(64 * WAVES) probe(const float* W, float* out, int heads, unsigned long long* clocks) {
    __shared__ __attribute__((aligned(16))) float At[64 * 68 + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 68 + 64; i += blockDim.x) At[i] = 0.001f * static_cast<float>((i * 37) & 255);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 3 * 256 * 256 * 4, 0x00020000);
    const int voff = (wave * 64 + lane) * 16;
    constexpr int NT = 16 / WAVES;  // column tiles per wave
    floatx4 acc[NT];
    for (int c = 0; c < NT; ++c) acc[c] = floatx4{0, 0, 0, 0};
    const float* ap = At + lane;
    unsigned long long t0 = 0, t1 = 0;
    if constexpr (WAVES == 4 && VAR == 0) {
        floatx4 Bq[kDepth];
        for (int i = 0; i < kDepth; ++i) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 4096, 0));
        float av[4];
        for (int i = 0; i < 4; ++i) av[i] = ap[i * 68];
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int h = 0; h < heads; ++h) {
            const int hb = (h % 3) * 262144;
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int sbase = hb + ((r + 1) & 3) * 65536 + (r == 3 ? 0 : 0);
                const float* apn = ap + (((r * 16 + 4) & 63)) * 68;
#pragma unroll
                for (int i = 0; i < kDepth; ++i) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i % 4], Bq[i][c], acc[c], 0, 0, 0);
                    Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + i * 4096, 0));
                    av[i % 4] = apn[(i % 12) * 68];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    } else if constexpr (WAVES == 4 && VAR != 0) {
        // VAR 1: A operands as one ds_read_b128 per FOUR k-steps; VAR 2: no A operand reads (register constant); VAR 3: no weight refills
        floatx4 Bq[kDepth];
        for (int i = 0; i < kDepth; ++i) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 4096, 0));
        const floatx4* ap4 = reinterpret_cast<const floatx4*>(At) + lane;
        floatx4 a4[2] = {ap4[0], ap4[64]};
        float a1 = At[lane];
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int h = 0; h < heads; ++h) {
            const int hb = (h % 3) * 262144;
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int sbase = hb + ((r + 1) & 3) * 65536;
#pragma unroll
                for (int i = 0; i < kDepth; ++i) {
                    const float aop = VAR == 1 ? a4[(i >> 2) & 1][i & 3] : a1;
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, Bq[i][c], acc[c], 0, 0, 0);
                    if (VAR != 3) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + i * 4096, 0));
                    if (VAR == 1 && (i & 3) == 3) a4[(i >> 2) & 1] = ap4[64 * (((i >> 2) + 2 + 4 * r) & 15)];
                    if (VAR == 3) a1 = At[lane + 68 * (i & 7)];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    } else {
        // 8 waves: wave w owns columns 32 w .. 32 w + 31 = 2 tiles; one dwordx4 = the B operands of TWO k-steps (2 tiles x 2 k-steps)
        floatx4 Bq[kDepth / 2];
        for (int i = 0; i < kDepth / 2; ++i) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 8192, 0));
        float av[4];
        for (int i = 0; i < 4; ++i) av[i] = ap[i * 68];
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int h = 0; h < heads; ++h) {
            const int hb = (h % 3) * 262144;
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int sbase = hb + ((r + 1) & 3) * 65536;
                const float* apn = ap + (((r * 16 + 4) & 63)) * 68;
#pragma unroll
                for (int i = 0; i < kDepth; ++i) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i % 4], Bq[i / 2][2 * (i & 1)], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i % 4], Bq[i / 2][2 * (i & 1) + 1], acc[1], 0, 0, 0);
                    if (i & 1) Bq[i / 2] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + (i / 2) * 8192, 0));
                    av[i % 4] = apn[(i % 12) * 68];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    float s = 0.f;
    for (int c = 0; c < NT; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) clocks[blockIdx.x * 8 + wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int heads = argc > 1 ? atoi(argv[1]) : 300, grid = argc > 2 ? atoi(argv[2]) : 256;
    float *W, *out;
    unsigned long long* clk;
    HIP_OK(hipMalloc(&W, 3 * 256 * 256 * 4));
    HIP_OK(hipMalloc(&out, grid * 512 * 4));
    HIP_OK(hipMalloc(&clk, grid * 8 * 8));
    std::vector<float> hw(3 * 256 * 256);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 1e-3f * static_cast<float>((i * 2654435761u) >> 24);
    HIP_OK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned long long> hc(grid * 8);
    for (int variant : {4, 8, 41, 42, 43, 4, 41}) {
        HIP_OK(hipMemset(clk, 0, grid * 8 * 8));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipEventRecord(e0));
        if (variant == 4) probe<4, 0><<<grid, 256>>>(W, out, heads, clk);
        else if (variant == 41) probe<4, 1><<<grid, 256>>>(W, out, heads, clk);
        else if (variant == 42) probe<4, 2><<<grid, 256>>>(W, out, heads, clk);
        else if (variant == 43) probe<4, 3><<<grid, 256>>>(W, out, heads, clk);
        else probe<8, 0><<<grid, 512>>>(W, out, heads, clk);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipDeviceSynchronize());
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        HIP_OK(hipMemcpy(hc.data(), clk, grid * 8 * 8, hipMemcpyDeviceToHost));
        std::vector<double> per;
        for (int i = 0; i < grid * 8; ++i) if (hc[i]) per.push_back(static_cast<double>(hc[i]) / heads);
        std::sort(per.begin(), per.end());
        printf("variant %2d (4 / 8 = waves per workgroup; 41 = A operands by ds_read_b128 per four k-steps, 42 = no A reads, 43 = no weight refills): %6.0f clocks per head (median), launch %.3f ms = %.2f us per head; floor 8192\n",
               variant, per[per.size() / 2], ms, 1e3 * ms / heads);
    }
    return 0;
}
