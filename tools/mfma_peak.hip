// Sustained v_mfma_f32_16x16x4_f32 rate of the whole chip from registers only (no memory traffic): the practical ceiling that
// the network kernels' "f32 MFMA" roofline fractions should be read against (the 157.3 TFLOP/s peak assumes 2.4 GHz).
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
    floatx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
        for (int ms_target : {2, 20, 200}) {
            const int grid = 256 * wgs_per_cu;
            const int iters = ms_target * 4500;  // 16 MFMAs x 32 cycles per iteration at ~2.3 GHz
            mfma_loop<16><<<grid, 256>>>(out, 1000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            mfma_loop<16><<<grid, 256>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * 16 * 16 * 4 * 16.0 * iters * 4.0 * grid;  // per wave per MFMA 2*16*16*4, 4 waves per WG
            printf("wgs/CU %d  %7.2f ms  %7.1f TFLOP/s  (implied MFMA clock %.2f GHz at 256 FLOP/clk/CU x 256 CUs)\n", wgs_per_cu, ms,
                   flops / ms / 1e9, flops / ms / 1e6 / (256.0 * 256.0 * (wgs_per_cu > 1 ? 1 : 1)) / 1e3);
        }
    return 0;
}
