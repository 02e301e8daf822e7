// graph_nested_fork.hip -- the minimal stream-capture sequence of round 2's sub-batch pipelining with a side stream per sub-batch:
// a SECOND level of fork / join (sub-batch stream -> its novelty-lookup stream) inside a captured step whose first level already forks
// the sub-batch streams from the capturing stream.  Round 2 saw `hipGraphInstantiate` take the host process down (gpurun_out/r2_t26.log,
// r2_b23.log) with such a capture and went back to in-line lookups; whether that was the runtime or a mis-joined stream in
// the uncommitted code could not be decided from the logs.  This program issues exactly that sequence with trivial kernels -- every forked
// stream is joined to the origin before hipStreamEndCapture, no captured event is waited on by a non-capturing stream -- so ONE run
// on an MI355X box decides it:  hipcc --offload-arch=gfx950 -o tools/bin/graph_nested_fork tools/graph_nested_fork.hip && tools/bin/graph_nested_fork
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define OK(e)                                                                                         \
    do {                                                                                              \
        hipError_t r_ = (e);                                                                          \
        if (r_ != hipSuccess) { std::printf("FAILED %s: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); return 2; } \
    } while (0)

__global__ void bump(int* p, int v) { atomicAdd(p + (threadIdx.x & 7), v); }

int main(int argc, char** argv) {
    const int shards = argc > 1 ? std::atoi(argv[1]) : 4, sims = argc > 2 ? std::atoi(argv[2]) : 8;
    int* d = nullptr;
    OK(hipMalloc(&d, 64));
    OK(hipMemset(d, 0, 64));
    hipStream_t origin, sh[8], side[8];
    hipEvent_t fork1, join1[8], fork2[8], join2[8];
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
    OK(hipEventRecord(fork1, origin));                                            // level 1 fork: sub-batch streams
    for (int k = 1; k < shards; ++k) OK(hipStreamWaitEvent(sh[k], fork1, 0));
    for (int s = 0; s < sims; ++s)
        for (int k = 0; k < shards; ++k) {
            hipStream_t st = k == 0 ? origin : sh[k];                             // sub-batch 0 runs on the capturing stream itself
            bump<<<1, 64, 0, st>>>(d, 1);                                         // tree + env step
            OK(hipEventRecord(fork2[k], st));                                     // level 2 fork: this sub-batch's lookup stream
            OK(hipStreamWaitEvent(side[k], fork2[k], 0));
            bump<<<1, 64, 0, side[k]>>>(d, 100);                                  // novelty lookup
            OK(hipEventRecord(join2[k], side[k]));
            bump<<<1, 64, 0, st>>>(d, 10000);                                     // forward
            OK(hipStreamWaitEvent(st, join2[k], 0));                              // level 2 join
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
