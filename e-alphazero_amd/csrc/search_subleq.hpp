// search_subleq.hpp -- subleq16_search_kernel: every simulation of a Subleq-16 search in ONE launch on GLOBAL-MEMORY trees: round 2's C3
// path.  Since round 3 search_subleq_persist.hpp runs both self-play and eaz_search_run on LDS trees; this kernel remains for trees
// beyond that kernel's LDS limit (S > 68) and as the EAZ_TUNE_PERSIST = 2 variant the tests compare against.
//
// Reference: the simulation loop of emctx [EXT] search as called from src/selfplay.py:107-117, with the recurrent fn
// of src/context.py:117-155 (env.step of src/envs/subleq.py + forward of src/network/fully_connected.py:41-101).
//
// One workgroup of 512 threads (8 waves) owns 16 trees for all S simulations.  Per simulation it runs the three device functions the
// per-simulation path launches as kernels, over its own 16 trees only, separated by workgroup barriers:
//   T  tree_step_body<16, 1>  (tree.hpp): expand + backward of simulation s-1, select of simulation s; 16 lanes per tree, 4 trees
//      per wave; the trees stay in global memory (133 words per node at A = 16: 16 trees x 65 nodes do not fit in LDS) but no other
//      workgroup touches them, so they are served from this CU's L1 / the XCD's L2;
//   E  subleq16_wave_body     (env_kernels.hpp): env.step of the chosen edges, one environment per wavefront at a time, claimed from a
//      work queue by the 8 waves; the 48 row-words of each new state stay in LDS;
//   F  fc16p_body             (net_fc_pipe.hpp): the forward of the three heads self-play reads for the 16 new observations (one f32 tile
//      built from the row-words), one head per group of four waves at a time, two side by side; the group without a head in the second
//      round looks the observations' novelty up.
// Same arithmetic, same order as the per-simulation kernels (parity suites run both).  What changes is the schedule:
//   * no launches inside a search (the per-simulation path pays two dependent graph edges of ~9 us per simulation);
//   * a slow interpreter run holds up its own 16 trees for one simulation, not the whole batch: the per-simulation launch waits for
//     the slowest of 4096 programs (a few per thousand run all 200 cycles) in EVERY simulation;
//   * the price: the MFMA pipes of a CU idle while its workgroup is in phases T and E.
// Visibility between phases: global stores of one wave are read by other waves of the same workgroup after fence + barrier (one CU,
// one L1: workgroup scope).
#pragma once
#include "net_fc_pipe.hpp"
#include "search_kernels.hpp"

namespace eaz {

struct Sq16SearchArgs {
    TreeDev t;
    EnvDev e;
    NetDev net;
    NetOut out;
    HeadList heads;
    uint8_t* obs;         // [B, 240] observation rows of the simulation in flight
    float* next_reward;   // [B]
    uint8_t* next_term;   // [B]
    uint8_t* next_legal;  // [B, 16]
    NoveltyFuse nf;
    int sim0, n_sims;     // simulations sim0 .. sim0 + n_sims - 1
    int expand_first;     // the outputs of simulation sim0 - 1 are pending (stepwise caller)
    int expand_last;      // also expand + back up the last simulation (otherwise left pending)
    unsigned long long* stamps;  // diagnostic build: [256][32] shader-clock stamps of simulation n_sims / 2 (tools/subleq_stamps.py)
};

#ifdef EAZ_FC_STAMPS
#define EAZ_QSTAMP(i)                                                                                  \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (dg.stamps && s == dg.n_sims / 2 && (tid & 63) == 0 && tid < 256 && blockIdx.x < 256) {     \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            dg.stamps[blockIdx.x * 32 + (tid >> 6) * 8 + (i)] = t_;                                    \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#define EAZ_ESTAMP(i)                                                                                  \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (dg.stamps && s == dg.n_sims / 2 && (tid & 63) == 0 && blockIdx.x < 256) {                  \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            dg.stamps[256 * 32 + blockIdx.x * 32 + (tid >> 6) * 4 + (i)] = t_;                         \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#else
#define EAZ_QSTAMP(i)
#define EAZ_ESTAMP(i)
#endif

constexpr int kSqWaves = 8;     // 2 per SIMD (<= 256 registers each): T on waves 0-3, E on all, F = one head per 4 waves at a time
constexpr int kSqThreads = 64 * kSqWaves;
constexpr int kSqHeadGroups = kSqWaves / 4;

// The kernel's arguments live in the kernarg segment and every phase reads them through a pointer the compiler cannot see through
// (an empty asm redefines it per phase): a by-value struct has all its loads hoisted to the kernel entry as loop invariants, and
// the ~250 SGPRs the three phases need between them then spill through VGPR lanes into scratch memory -- inside the latency chains of
// the tree walk (measured: tree phase 28 k -> 53 k cycles).  Re-issued scalar loads hit the scalar cache.
// (Sq16SearchArgs is the kernel's ONLY parameter, so it sits at offset 0 of the kernarg segment.)
typedef const Sq16SearchArgs* SqArgsPtr;
__device__ __forceinline__ SqArgsPtr sq_args() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const Sq16SearchArgs*)p;  // constant -> generic; the loads are inferred back to scalar loads after inlining
}

__global__ void __launch_bounds__(kSqThreads) subleq16_search_kernel(Sq16SearchArgs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ int sel_node[16], sel_action[16], sel_dst[16];
    __shared__ int env_next;  // environments kSqWaves + env_next .. 15 of the simulation in flight are not claimed yet
    __shared__ __attribute__((aligned(16))) uint8_t sq_words[16][48];  // row-words of the 16 new states, 5 observation bits each
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * 16;
    // every phase derives its lane roles from a thread id the compiler cannot trace to threadIdx: otherwise the address arithmetic and
    // lane masks of all three phases are hoisted out of the simulation loop as invariants and held in registers across it
    auto phase_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    int n_sims, sim0, expand_first, expand_last;
    {
        SqArgsPtr a = sq_args();
        n_sims = a->n_sims; sim0 = a->sim0; expand_first = a->expand_first; expand_last = a->expand_last;
    }
    {
        // inputs 240 .. 255 of the shared observation tile stay zero for the whole launch (layer 1 runs whole rounds of 16 k-steps)
        float* Xs = reinterpret_cast<float*>(lds + kSqHeadGroups * fp_lds_bytes(240, 16));
        for (int idx = tid; idx < 16 * 16; idx += kSqThreads) Xs[h16_off(240 + (idx >> 4), idx & 15)] = 0.0f;
    }
#ifdef EAZ_FC_STAMPS
    struct { unsigned long long* stamps; int n_sims; } dg{sq_args()->stamps, n_sims};  // what EAZ_QSTAMP names
#endif
#pragma unroll 1
    for (int s = 0; s < n_sims; ++s) {
        const int sim = sim0 + s;
        const int mode = kTreeSelect | ((s > 0 || expand_first) ? kTreeExpand : 0);
        // ---- T ----
        EAZ_QSTAMP(0);
        if (tid == 0) env_next = 0;  // nobody is in the interpreter phase: between the barrier that ended F and the one that ends T
        if (wave < 4) {
            SqArgsPtr a = sq_args();
            const int pt = phase_tid(), sub = pt & 15;
            const TreeSel sel = tree_step_body<16, 1>(a->t, sim, mode, b0 + (pt >> 4), sub, b0);
            EAZ_QSTAMP(1);
            if (sub == 0) { sel_node[pt >> 4] = sel.node; sel_action[pt >> 4] = sel.action; sel_dst[pt >> 4] = sel.dst; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
        // ---- E: the interpreters, one environment per wave at a time ----
        EAZ_QSTAMP(2);
        {
            SqArgsPtr a = sq_args();
            const int lane = phase_tid() & 63;
            const int B = a->t.B, N = a->t.N, sb = a->t.state_bytes;
            char* emb = a->t.embeddings;
            // every wave takes its own environment first, then whatever is left, one at a time: a few interpreter runs are 3x the median
            // and the phase ends with its last wave.  The claim is ONE LDS atomic per wave and environment, written as the instruction itself:
            // `atomicAdd` goes through the compiler's atomic optimizer, which turned a lane-0 form into a loop that never ended and an
            // all-lane form into a 64-iteration scan behind `s_waitcnt vmcnt(0)` (+6 k cycles per claim).
            EAZ_ESTAMP(0);
            {
                // the parent states of two of the workgroup's environments: touched now, all 16 in flight together across the 8 waves, so that
                // whichever wave interprets one starts from L2 / L1 instead of paying a memory round trip in turn
                int touch = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int jt = wave + kSqWaves * k;
                    const bool in_batch = b0 + jt < B;  // a tree beyond the batch: its selection is tree 0's (tree_step_body), never a real edge
                    const char* spt = emb + (static_cast<long>(in_batch ? b0 + jt : B - 1) * N + (in_batch ? sel_node[jt] : 0)) * sb;
                    if (lane < 24) touch ^= reinterpret_cast<const int*>(spt)[lane];
                }
                asm volatile("" ::"v"(touch));
            }
            int j = wave, taken = 0;
            (void)taken;  // read by the diagnostic build's stamps only
            const uint32_t env_next_off = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) int*)(&env_next)));
#pragma unroll 1
            while (j < 16) {
                const int b = b0 + j;
                if (b < B) {
                    const int node = __builtin_amdgcn_readfirstlane(sel_node[j]), action = __builtin_amdgcn_readfirstlane(sel_action[j]);
                    const int dst = __builtin_amdgcn_readfirstlane(sel_dst[j]);
                    char* base = emb + static_cast<long>(b) * N * sb;
                    subleq16_wave_body(a->e, base + static_cast<long>(node) * sb, base + static_cast<long>(dst) * sb, b, lane, action, 0, Key{0, 0},
                                       nullptr, nullptr, a->next_reward, a->next_term, a->next_legal, 16, nullptr, a->nf, &sq_words[j][0]);
                }
                taken += 1;
                EAZ_ESTAMP(taken < 3 ? taken : 3);
                j = kSqWaves + lds_claim_one(env_next_off);  // one lane, one ds_add_rtn_u32 (search_persist.hpp)
            }
        }
        EAZ_QSTAMP(3);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
        // ---- F: one head per group of four waves at a time, side by side (their barriers coincide) ----
        EAZ_QSTAMP(4);
        {
            SqArgsPtr a = sq_args();
            const int ft = phase_tid();
            const int hg = wave >> 2, pt = ft & 255;
            // the 16 new observations as ONE f32 tile for all three heads, straight from the row-words the interpreters left in LDS
            // (element (5 w + j, row) = bit j of word w: subleq.py:88-98); the first barrier of the bodies below publishes it.  The
            // observation rows never go through memory.
            float* Xs = reinterpret_cast<float*>(lds + kSqHeadGroups * fp_lds_bytes(240, 16));
            const int32_t* inv = a->net.in_inv;  // layer-1 input order (eaz_net_set_input_order): input k sits at position inv[k] of the chain
            for (int idx = ft; idx < 16 * 48; idx += kSqThreads) {
                const int row = idx / 48, w = idx - row * 48;
                const uint32_t code = sq_words[row][w];
#pragma unroll
                for (int j = 0; j < 5; ++j) Xs[h16_off(inv ? inv[5 * w + j] : 5 * w + j, row)] = (code >> j) & 1u ? 1.0f : 0.0f;
            }
#pragma unroll 1
            for (int h0 = 0; h0 < a->heads.n; h0 += kSqHeadGroups) {
                if (h0 + hg < a->heads.n) {
                    fc16p_body(a->net, a->obs, a->t.B, a->out, a->heads.id[h0 + hg], b0, lds + hg * fp_lds_bytes(240, 16), pt, nullptr, Xs);
                } else {
                    // this head group has no head left: one of its waves looks the 16 observations' novelty up meanwhile (XXHash variant,
                    // hashes.py:162-229: lane = (environment, stripe); the IO rows or the whole observation, sq16_words_hash_index)
                    if ((wave & 3) == 0) {
                        const int lane = ft & 63, e = lane >> 2, i = lane & 3;
                        const uint32_t index = sq16_words_hash_index(&sq_words[e][0], i, a->net.hash_off == 0, a->nf.hash_bits);
                        if (i == 0 && b0 + e < a->t.B) subleq16_novelty(a->nf, b0 + e, index);
                    }
#pragma unroll
                    for (int k = 0; k < kFpBarriers; ++k) __syncthreads();  // the barriers of fc16p_body
                }
            }
        }
        EAZ_QSTAMP(5);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
        EAZ_QSTAMP(6);
    }
    if (expand_last && wave < 4) {
        SqArgsPtr a = sq_args();
        const int pt = phase_tid();
        tree_step_body<16, 1>(a->t, sim0 + n_sims, kTreeExpand, b0 + (pt >> 4), pt & 15, b0);
    }
}

}  // namespace eaz
