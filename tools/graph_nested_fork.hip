// graph_nested_fork.hip -- the minimal stream-capture sequence of the sub-batch pipelining with a side stream per sub-batch: a SECOND level
// of fork / join (sub-batch stream -> its novelty-lookup stream) inside a captured step whose first level already forks the sub-batch
// streams from the capturing stream.  Every forked stream is joined to the origin before hipStreamEndCapture and no captured event is waited
// on by a non-capturing stream: the capture is legal.  Measured on MI355X (round 3):
//   * linked against /opt/rocm-7.2.0's libamdhip64.so.7 (the default of this build line): captured, instantiated, replayed, sums correct for
//     every size tried (2..8 sub-batches x 1..200 simulations, all variants);
//   * with PyTorch 2.10's bundled runtime (libamdhip64.so 7.0.2; `LD_LIBRARY_PATH=$T LD_PRELOAD=$T/libamdhip64.so`, T = torch/lib) -- the
//     runtime libeaz.so has to share with torch in one process -- hipStreamEndCapture segfaults as soon as there are TWO sub-batches, even
//     with one simulation (`graph_nested_fork 2 1`); one sub-batch, i.e. a single fork level, works at any size (`1 64 1`).  rocgdb on the
//     engine's own capture shows hip::Stream::EndCapture() calling itself until the stack is gone.
// That is the host crash of round 2 (gpurun_out/r2_t26.log, r2_b23.log), so the engine keeps one fork level (eaz_api.hip make_shard).
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/graph_nested_fork tools/graph_nested_fork.hip && tools/bin/graph_nested_fork [sub-batches] [simulations] [variant bits]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define OK(e)                                                                                         \
    do {                                                                                              \
        hipError_t r_ = (e);                                                                          \
        if (r_ != hipSuccess) { std::printf("FAILED %s: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); return 2; } \
    } while (0)

__global__ void bump(int* p, int v) { atomicAdd(p + (threadIdx.x & 7), v); }
struct Big { int pad[120]; int* p; int v; };  // a by-value argument struct of the size the engine's kernels take (TreeDev, NetDev)
__global__ void bump_big(Big b) { atomicAdd(b.p + (threadIdx.x & 7), b.v); }

int main(int argc, char** argv) {
    const int shards = argc > 1 ? std::atoi(argv[1]) : 4, sims = argc > 2 ? std::atoi(argv[2]) : 8;
    // bit 0: the capturing stream forks / joins a side stream of its own BEFORE the sub-batch part (the root inference of the real step);
    // bit 1: kernels with large by-value argument structs; bit 2: sub-batch 0 (which runs on the capturing stream) has no side stream
    const int variant = argc > 3 ? std::atoi(argv[3]) : 0;
    int* d = nullptr;
    OK(hipMalloc(&d, 64));
    OK(hipMemset(d, 0, 64));
    hipStream_t origin, pside, sh[8], side[8];
    hipEvent_t fork1, join1[8], fork2[8], join2[8], pf, pj;
    OK(hipStreamCreateWithFlags(&pside, hipStreamNonBlocking));
    OK(hipEventCreateWithFlags(&pf, hipEventDisableTiming));
    OK(hipEventCreateWithFlags(&pj, hipEventDisableTiming));
    int extra = 0;
    Big big{};
    big.p = d;
    OK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
    OK(hipEventCreateWithFlags(&fork1, hipEventDisableTiming));
    for (int k = 0; k < shards; ++k) {
        OK(hipStreamCreateWithFlags(&sh[k], hipStreamNonBlocking));
        OK(hipStreamCreateWithFlags(&side[k], hipStreamNonBlocking));
        OK(hipEventCreateWithFlags(&join1[k], hipEventDisableTiming));
        OK(hipEventCreateWithFlags(&fork2[k], hipEventDisableTiming));
        OK(hipEventCreateWithFlags(&join2[k], hipEventDisableTiming));
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    OK(hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal));
    if (variant & 1) {
        bump<<<1, 64, 0, origin>>>(d, 0);
        OK(hipEventRecord(pf, origin));
        OK(hipStreamWaitEvent(pside, pf, 0));
        bump<<<1, 64, 0, pside>>>(d, 0);
        OK(hipEventRecord(pj, pside));
        bump<<<1, 64, 0, origin>>>(d, 0);
        OK(hipStreamWaitEvent(origin, pj, 0));
        bump<<<1, 64, 0, origin>>>(d, 0);
        extra = 0;
    }
    OK(hipEventRecord(fork1, origin));                                            // level 1 fork: sub-batch streams
    for (int k = 1; k < shards; ++k) OK(hipStreamWaitEvent(sh[k], fork1, 0));
    for (int s = 0; s < sims; ++s)
        for (int k = 0; k < shards; ++k) {
            hipStream_t st = k == 0 ? origin : sh[k];                             // sub-batch 0 runs on the capturing stream itself
            if (variant & 2) { big.v = 1; bump_big<<<1, 64, 0, st>>>(big); }
            else bump<<<1, 64, 0, st>>>(d, 1);                                    // tree + env step
            const bool has_side = !((variant & 4) && k == 0);
            if (has_side) {
                OK(hipEventRecord(fork2[k], st));                                 // level 2 fork: this sub-batch's lookup stream
                OK(hipStreamWaitEvent(side[k], fork2[k], 0));
            }
            bump<<<1, 64, 0, has_side ? side[k] : st>>>(d, 100);                  // novelty lookup
            if (has_side) OK(hipEventRecord(join2[k], side[k]));
            if (variant & 2) { big.v = 10000; bump_big<<<1, 64, 0, st>>>(big); }
            else bump<<<1, 64, 0, st>>>(d, 10000);                                // forward
            if (has_side) OK(hipStreamWaitEvent(st, join2[k], 0));                // level 2 join
        }
    for (int k = 1; k < shards; ++k) {
        OK(hipEventRecord(join1[k], sh[k]));                                      // level 1 join
        OK(hipStreamWaitEvent(origin, join1[k], 0));
    }
    OK(hipStreamEndCapture(origin, &graph));
    std::printf("captured: %d sub-batches x %d simulations, two fork levels\n", shards, sims);
    std::fflush(stdout);
    OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    std::printf("instantiated\n");
    std::fflush(stdout);
    for (int i = 0; i < 3; ++i) OK(hipGraphLaunch(exec, origin));
    OK(hipStreamSynchronize(origin));
    int h[16];
    OK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
    const int want = 3 * shards * sims * 64 * (1 + 100 + 10000);  // 64 threads add v per launch
    int got = 0;
    for (int i = 0; i < 8; ++i) got += h[i];
    std::printf("%s: sum %d (expected %d)\n", got == want ? "REPLAY OK" : "WRONG RESULT", got, want);
    return got == want ? 0 : 1;
}
