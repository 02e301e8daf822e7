// search_persist.hpp -- the whole Gumbel-MuZero search of 16 trees in ONE persistent workgroup (DeepSea + FC network, two actions).
//
// Reference: emctx search loop [EXT] as called from src/selfplay.py:107-117, with the recurrent fn of
// src/context.py:117-155 (DeepSea env.step of src/envs/deep_sea.py:59-81, EpistemicFullyConnectedAZNet forward of
// src/network/fully_connected.py:41-101, XXHash novelty of src/network/hashes.py:162-229).  In the reference the S simulations are one
// `fori_loop` of one device program; here they are one kernel launch.
//
// Why one workgroup can own its trees for all S simulations: trees are independent, and the FC forward works on 16-row tiles -- a
// workgroup evaluates the network for exactly its own 16 leaves, so there is no hand-off between workgroups at all and the tree
// statistics never leave LDS (north-star: "one wavefront walking one tree in LDS-resident child-index/visit/prior arrays").
//
// Workgroup = 512 threads:
//   waves 0-3  tree phase: 16 lanes per tree (lane j = level j of the recorded root-to-leaf path); forward: each wave owns 64 output
//              columns of the hidden layers (v_mfma_f32_16x16x4_f32, one accumulation chain per output = the oracle's fmaf chain)
//   waves 4-7  helpers, one per SIMD: W1 row gathers (layer 1) of heads 1 and 2 under layer 2 of the head before (helper hw: rows 4 hw .. 4 hw + 3
//              of the H1 tile -- one helper doing all 16 rows cost its SIMD partner 0.9 k cycles per head, which the other three MFMA waves
//              then waited for at the head's barrier); wave 4 also: novelty lookup of the 16 leaves, and at the end ((c0 + c1) + (c2 + c3)) + b
//              and the head activations from the four quarter chains of layer 3, which waves 0-3 run as MFMA chains over their own hidden
//              columns (a head's chain rides in the first round of the next head's layer 2)
//
// Per simulation (all in LDS / registers, nothing but the weight stream and 16 W1 rows per head comes from L2):
//   expand + backward of the previous leaf   lane 0 writes the new node; the path's levels are folded leaf-to-root: every lane computes
//                                            the discounted return G_d of its own level (serial fmaf order of A.6), then the running-mean
//                                            updates of all levels are independent
//   re-score                                 selection inputs of a node change only when the node is on the backed-up path, so the argmax
//                                            of A.4 is re-evaluated for the D+1 nodes of the path in parallel (one lane per node) and cached
//                                            as (action, child) per node
//   select                                   the root-to-leaf walk is a pointer chase through the cached choices (one LDS read per level)
//   DeepSea env.step of the chosen edge, one-hot index of the child's observation
//   forward                                  layer 1 = row gather W1[hot] (exact for a one-hot row), layer 2 on MFMA with the W2 stream
//                                            prefetched 16 k-steps ahead in registers (the stream is cyclic over the three heads, so the
//                                            first k-steps of the next simulation are already in flight during the tree phase)
//
// Numerics: every float operation is the one tree_step_kernel / fc16_forward_kernel perform, in the same order (A = 2: the butterfly
// sum of two slots is x0 + x1), so the results are bit-identical to the per-simulation kernels and to the CPU oracle.
#pragma once
#include "net_fc.hpp"
#include "tree.hpp"

namespace eaz {

constexpr int kPsTrees = 16;     // trees per workgroup = rows of the forward tile
constexpr int kPsThreads = 512;  // 4 tree / MFMA waves + 4 helper waves (one per SIMD: the layer-1 work of the next head costs every MFMA wave a quarter
                                  // instead of wave 0 all of it -- 0.9 k cycles of waiting at each head's barrier, profiles/r03_i_persist_stamps.txt)
constexpr int kPsA = 2;          // actions (DeepSea)
#ifndef EAZ_PS_DEPTH
#define EAZ_PS_DEPTH 16
#endif
constexpr int kPsDepth = EAZ_PS_DEPTH;  // k-steps of the W2 stream in flight per wave (16 or 32; 4 VGPRs each)
constexpr int kPsARing = 8;             // A operands (LDS) read this many k-steps ahead
static_assert(64 % kPsDepth == 0 && 64 / kPsDepth >= 2 && kPsDepth % kPsARing == 0, "a head's 64 k-steps are >= 2 whole rounds of kPsDepth");

// node record in LDS (32-bit words)
enum : int {
    PR_RAW_VALUE = 0, PR_NODE_VALUE = 1, PR_RAW_VAR = 2, PR_NODE_VAR = 3, PR_NODE_VISITS = 4,
    PR_INDEX = 5, PR_VISITS = 7, PR_LOGITS = 9, PR_VALUES = 11, PR_REWARDS = 13, PR_DISCOUNTS = 15, PR_VAR = 17,
    PR_SEL = 19,    // cached selection: action << 16 | (child index & 0xFFFF)
    PR_STATE = 20,  // DeepSea state, packed (ps_pack)
    PR_WORDS = 21
};

// Fused self-play (src/selfplay.py:78-149): with `on` the launch runs T real steps of step_fn for its 16 environments -- root inference,
// root preparation (selfplay.py:92-106), the S simulations, final action / action weights (A.2), summary (A.7), action choice (:118-134),
// auto-reset env.step (:135) and SelfplayOutput[t] (:136-143) -- and never touches the global tree arrays.
enum : int { kSpUniform = 1, kSpSampleVisits = 2, kSpSampleImproved = 4 };
struct PersistSelfplay {
    int on, T, flags;
    Key key;                  // key of the call; key_seq[t] = fold(key, kTagStep + t) (selfplay.py:147)
    float sbeta, gumbel_scale;
    char* states;             // [B] DeepSea states: last_states in, new last_states out
    const float* gumbel;      // explicit unscaled root noise [T,B,A] or null (generated from the step's key)
    char* out_states;         // SelfplayOutput fields [T,B(,A)]; any may be null
    float *out_root_value, *out_root_std, *out_value_pred, *out_ube_pred, *out_qvar;
    int32_t* out_action;
    // on == 2: the evaluate driver (src/evaluate.py:13-57) as a device-side loop: T = max_episode_length + 1 is a cap, sbeta the constant
    // exploitation beta (:34), the real step is a plain env.step (:47, no auto-reset); a workgroup stops when its 16 environments have all
    // terminated (the reference's loop runs on until EVERY environment has: for an environment that is already terminal those further steps
    // only zero its rewards field, which eval_finalize does from the workgroups' iteration counts)
    float* eval_sum;          // [B] sum of rewards per environment (:48-50)
    int* eval_iters;          // [1] max over the workgroups of their iteration counts = the trip count of the reference's while_loop
    int* eval_wg_iters;       // [workgroups] iterations of each workgroup
};

struct PersistArgs {
    PersistSelfplay sp;
    TreeDev t;       // options + the global SoA arrays (root in, whole tree out)
    NetDev net;
    EnvDev env;
    int head[3];     // value, ube, policy head ids (0, 1, 2|3)
    int amap_lds;    // action map staged in LDS (G*G <= kPsAmapMax)
    unsigned long long* stamps;  // diagnostic builds only
};

constexpr int kPsAmapMax = 4096;
constexpr int kPsMiscBytes = 3584;

struct PersistLds {
    int tstride;                      // words per tree
    int off_tree, off_h1, off_h2, off_part, off_w3, off_path, off_misc, off_amap, off_table, total;  // byte offsets
};

__host__ __device__ inline PersistLds persist_lds(int N, int S, int G) {
    PersistLds L;
    L.tstride = (N * PR_WORDS) | 1;  // odd: trees start in different banks
    int o = 0;
    L.off_h1 = o; o += 2 * kF16HBytes;  // two H1 tiles: the helper fills the next head's while layer 2 reads the current one
    L.off_h2 = o; o += kF16HBytes;
    L.off_part = o; o += 3 * 4 * kPsTrees * 4 * 4;  // layer-3 partial sums [3 heads][4 waves][16 rows][4 outputs]
    L.off_w3 = o; o += 3 * kFcH * 4 * 4;
    L.off_misc = o; o += kPsMiscBytes;
    L.off_table = o; o += 3 * S * 4;
    L.off_path = o; o += ((kPsTrees * S * 2) + 15) & ~15;
    L.off_amap = o; o += (G * G <= kPsAmapMax) ? ((G * G + 15) & ~15) : 0;
    L.off_tree = o; o += kPsTrees * L.tstride * 4;
    L.total = (o + 15) & ~15;
    return L;
}

// per-tree scalars and forward outputs (LDS, inside the misc block)
struct PersistMisc {
    int D[kPsTrees], parent[kPsTrees], action[kPsTrees], dst[kPsTrees], hot[kPsTrees];
    float o_value[kPsTrees], o_ube[kPsTrees], o_nov[kPsTrees], o_reward[kPsTrees];
    float o_logit[kPsTrees][kPsA];
    int o_term[kPsTrees];
    float beta[kPsTrees];
    float gumbel[kPsTrees][kPsA];
    int invalid[kPsTrees][kPsA];
    float edge[kPsTrees][16][2];  // backward: (reward, discount) of the 16 path levels of the chunk being folded
    float root_v[kPsTrees], root_u[kPsTrees];  // fused self-play: value / ube prediction of the root (SelfplayOutput)
    uint32_t root_state[kPsTrees];             // fused self-play: the environment's current state, packed
    float sum_reward[kPsTrees];                // evaluate mode (PersistSelfplay::on == 2): sum of the real steps' rewards (evaluate.py:48-50)
    int real_term[kPsTrees];                   // evaluate mode: `terminated` of the environments' current states (1 for trees beyond the batch)
};
static_assert(sizeof(PersistMisc) <= kPsMiscBytes, "misc block too small");

// DeepSea state <-> one word: step_count (12 bits) | pos (12) | terminated | truncated | reward == 1
__device__ __forceinline__ uint32_t ps_pack(const eaz_deepsea_state& s) {
    return static_cast<uint32_t>(s.step_count & 0xFFF) | (static_cast<uint32_t>(s.pos & 0xFFF) << 12) |
           (static_cast<uint32_t>(s.terminated != 0) << 24) | (static_cast<uint32_t>(s.truncated != 0) << 25) |
           (static_cast<uint32_t>(s.reward == 1.0f) << 26);
}
__device__ __forceinline__ eaz_deepsea_state ps_unpack(uint32_t w) {
    eaz_deepsea_state s;
    s.step_count = static_cast<int>(w & 0xFFF);
    s.pos = static_cast<int>((w >> 12) & 0xFFF);
    s.terminated = static_cast<uint8_t>((w >> 24) & 1);
    s.truncated = static_cast<uint8_t>((w >> 25) & 1);
    s.reward = ((w >> 26) & 1) ? 1.0f : 0.0f;
    s.pad[0] = s.pad[1] = 0;
    return s;
}

// The packed state holds what the environment itself can produce for G < 4096 (deep_sea.py:54-85): step count and column in 12 bits each,
// the two flags as bits, the reward as "is 1.0".  A caller's hand-made blob may hold anything; the host runs this check on the blobs a call
// starts from and sends a batch with any misfit to the per-simulation kernels, which keep whole blobs (the Subleq-16 kernel does the same:
// sq16_roots_fit_kernel).
__global__ void __launch_bounds__(256) ds_roots_fit_kernel(const char* __restrict__ states, int B, int G, int* misfit, int epoch) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const uint4 w = *reinterpret_cast<const uint4*>(states + static_cast<long>(b) * sizeof(eaz_deepsea_state));
    const int step = static_cast<int>(w.x), pos = static_cast<int>(w.y);
    const bool ok = step >= 0 && step <= 0xFFF && pos >= 0 && pos < G && (w.z == 0u || w.z == 0x3F800000u) &&  // a column on the grid; reward +0.0f or 1.0f
                    (w.w & 0x0000FEFEu) == 0u;  // flags 0 / 1 (the two padding bytes are nobody's: every step kernel writes them as zero)
    if (!ok) *misfit = epoch;  // `misfit` is host memory mapped into the device's address space; the host compares it with this call's epoch
}

// LDS traffic between the lanes of one wavefront: DS operations of a wave execute in issue order, so only the compiler has to be
// kept from reordering them
// Work-queue claim of the one-launch Subleq kernels: `counter` (an LDS address) += 1 by ONE lane, returns the value before the add to the
// whole wavefront.  Written as the instructions themselves: `atomicAdd` goes through the compiler's atomic optimizer, which turned the
// lane-0 form into a loop that never ended in the release build and the all-lane form into a 64-iteration scan behind vmcnt(0) (DESIGN.md
// 4.4).  exec is narrowed to lane 0 around the ds_add_rtn_u32, so the claim does not depend on how the LDS orders the 64 same-address adds
// of one wavefront against another wavefront's (rounds 2-3 issued it from all 64 lanes and divided by 64).  Call with all lanes active.
__device__ __forceinline__ int lds_claim_one(const uint32_t counter_lds_addr) {
    int got;
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %1"
                 : "=&v"(got), "=&s"(saved)
                 : "v"(counter_lds_addr), "v"(1)
                 : "memory");
    return __builtin_amdgcn_readfirstlane(got);
}

__device__ __forceinline__ void ps_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef EAZ_FC_STAMPS
#define EAZ_PSTAMP(i)                                                                                  \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (a.stamps && (threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x < 256 && sim == a.t.S / 2) { \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            a.stamps[blockIdx.x * 32 + (threadIdx.x ? 16 : 0) + (i)] = t_;                             \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#else
#define EAZ_PSTAMP(i)
#endif

// pgx.Env.step (A.1) around deep_sea.py:59-81 on a state in registers; `flip` = action_map[row][pos] != 0 is fetched by the caller
// (LDS or global).  Identical to envs.hpp ds_step.
__device__ __forceinline__ int ps_ds_row(const eaz_deepsea_state& s, int G) {
    int rr = s.step_count;  // _step_count is incremented before _step: the row read is the pre-increment count (deep_sea.py:62)
    return rr < 0 ? 0 : (rr > G - 1 ? G - 1 : rr);
}
__device__ __forceinline__ eaz_deepsea_state ps_ds_step(eaz_deepsea_state s, int act, int flip, int G) {
    if (s.terminated | s.truncated) {
        s.reward = 0.0f;  // finished episode: no-op with zero rewards
        return s;
    }
    s.step_count += 1;
    const int shift = ((act == 0) ^ (flip != 0)) ? -1 : 1;        // :63
    int pos = s.pos + shift;
    pos = pos < 0 ? 0 : (pos > G - 1 ? G - 1 : pos);              // :64
    const int term = s.step_count >= G - 1;                       // :72
    s.pos = pos;
    s.terminated = static_cast<uint8_t>(term);
    s.reward = (term && pos == G - 1) ? 1.0f : 0.0f;              // :74-78
    return s;
}

// mctx seq_halving.score_considered + masked_argmax [EXT] (A.2) with the root noise / mask of this tree in registers
__device__ __forceinline__ int ps_root_pick(const Group<1, kPsA>& g, const NodeRows<kPsA>& r, const float (&cq)[kPsA], float considered_visit,
                                            const float (&gum)[kPsA], const int (&inv)[kPsA]) {
    const float m = g.vmax(r.logits);
    float sc[kPsA];
#pragma unroll
    for (int j = 0; j < kPsA; ++j) {
        float s = (gum[j] + (r.logits[j] - m)) + cq[j];
        s = s > -1e9f ? s : -1e9f;
        s = s + (static_cast<float>(r.visits[j]) == considered_visit ? 0.0f : -kInf);
        sc[j] = inv[j] ? -kInf : s;
    }
    return g.argmax(sc);
}

// One round of layer 2: kPsDepth k-steps of four MFMAs each on the prefetched weights Bq, every k-step followed by the refill of its
// registers (buffer load, k-step in the scalar offset) and of its A-operand slot.  CHAIN: the round also carries the 16 dependent MFMAs
// of the previous head's layer-3 quarter chain, one per k-step, operands kChRing k-steps ahead.
constexpr int kChRing = 4;
template <bool CHAIN>
__device__ __forceinline__ void ps_round(floatx4 (&acc)[4], floatx4 (&Bq)[kPsDepth], float (&av)[kPsARing], const float* apn,
                                         __amdgpu_buffer_rsrc_t rs, int voff, int sbase, floatx4& a3, const float* hpA, const float* wpB) {
    float cha[kChRing], chb[kChRing];
    if constexpr (CHAIN) {
#pragma unroll
        for (int q = 0; q < kChRing; ++q) { cha[q] = hpA[68 * q]; chb[q] = wpB[4 * q]; }
    }
#pragma unroll
    for (int i = 0; i < kPsDepth; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i % kPsARing], Bq[i][c], acc[c], 0, 0, 0);
        if constexpr (CHAIN) {
            if (i < 16) {
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(cha[i % kChRing], chb[i % kChRing], a3, 0, 0, 0);
                if (i + kChRing < 16) { cha[i % kChRing] = hpA[68 * (i + kChRing)]; chb[i % kChRing] = wpB[4 * (i + kChRing)]; }
            }
        }
        Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + i * 4096, 0));
        av[i % kPsARing] = apn[i * 68];
        __builtin_amdgcn_sched_barrier(0);  // keep every refill right behind the k-step that freed its registers
    }
}

// EVAL: the evaluate mode (PersistSelfplay::on == 2) is its own instantiation, so that self-play and search pay nothing for it (as a run-time
// flag it cost the C2 launch 1.8 %: 34.1 -> 34.7 ms)
template <bool EVAL>
__global__ void __launch_bounds__(kPsThreads) search_persist_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const TreeDev& t = a.t;
    const int N = t.N, S = t.S, B = t.B;
    const PersistLds L = persist_lds(N, S, a.env.G);
    float* H1 = reinterpret_cast<float*>(lds + L.off_h1);
    float* H2 = reinterpret_cast<float*>(lds + L.off_h2);   // every wave reads back only the 64 columns it wrote
    float* part = reinterpret_cast<float*>(lds + L.off_part);
    float* W3s = reinterpret_cast<float*>(lds + L.off_w3);  // [3 heads][4 outputs][256 k]
    PersistMisc& M = *reinterpret_cast<PersistMisc*>(lds + L.off_misc);
    int* tab = reinterpret_cast<int*>(lds + L.off_table);   // [3][S]
    uint16_t* path = reinterpret_cast<uint16_t*>(lds + L.off_path);  // [16][S]: node | action << 15
    uint8_t* amap = lds + L.off_amap;
    uint32_t* trees = reinterpret_cast<uint32_t*>(lds + L.off_tree);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool helper = wave >= 4;
    const int hw = wave - 4;  // helper: rows 4 hw .. 4 hw + 3 of the next head's H1 tile
    const int tree0 = blockIdx.x * kPsTrees;
    // head order rotated per workgroup: at any moment the CUs of an XCD stream three different W2 matrices from L2 instead of all
    // hammering the same lines (the order in which the heads are evaluated does not enter any result)
    int hidv[3];
#pragma unroll
    for (int h = 0; h < 3; ++h) hidv[h] = a.head[(h + blockIdx.x) % 3];

    // ---------------------------------------------------------------- prologue ----------------------------------------------------
    for (int i = tid; i < kPsTrees * L.tstride; i += kPsThreads) trees[i] = 0u;
    for (int i = tid; i < 3 * S; i += kPsThreads) {
        const int row = i / S;  // considered-visit sequences for 0, 1, 2 considered actions (rows beyond max_considered are never read)
        tab[i] = row <= t.max_considered ? t.table[i] : 0;
    }
    if (a.amap_lds)
        for (int i = tid; i < a.env.G * a.env.G; i += kPsThreads) amap[i] = a.env.action_map[i];
    for (int i = tid; i < 3 * kFcH; i += kPsThreads) {
        const int h = i / kFcH, k = i - h * kFcH;
        const int hid = hidv[h], No = hid < 2 ? 1 : kPsA;
        const float* W3 = a.net.w[hid * 3 + 2];
#pragma unroll
        for (int c = 0; c < 4; ++c) W3s[(h * 4 + c) * kFcH + k] = c < No ? W3[k * No + c] : 0.0f;
    }
    const bool fused = a.sp.on != 0;
    constexpr bool evaluate = EVAL;  // src/evaluate.py:13-57 as a device-side loop (PersistSelfplay::on == 2)
    if (evaluate && tid < kPsTrees) {
        M.sum_reward[tid] = 0.0f;
        M.real_term[tid] = tree0 + tid < B ? ds_load(a.sp.states + static_cast<long>(tree0 + tid) * sizeof(eaz_deepsea_state)).terminated != 0 : 1;
    }
    __syncthreads();
    // evaluate: cond_fn (evaluate.py:18-20) for this workgroup's environments -- every thread reads the same 16 flags behind a barrier
    auto all_terminated = [&]() {
        bool all = true;
        for (int j = 0; j < kPsTrees; ++j) all = all && M.real_term[j] != 0;
        return all;
    };
    const bool ev_skip = evaluate && (all_terminated() || a.sp.T <= 0);  // nothing to do: trip count 0
    if (tid < kPsTrees && !fused) {
        const int i = tid, b = tree0 + i < B ? tree0 + i : B - 1;  // trees beyond the batch mirror the last one and are never written back
        uint32_t* R = trees + i * L.tstride;
        for (int n = 0; n < N; ++n) {
            R[n * PR_WORDS + PR_INDEX] = 0xFFFFFFFFu;
            R[n * PR_WORDS + PR_INDEX + 1] = 0xFFFFFFFFu;
        }
        const long nb = static_cast<long>(b) * N;
        R[PR_RAW_VALUE] = f2u(t.raw_values[nb]);
        R[PR_NODE_VALUE] = f2u(t.node_values[nb]);
        R[PR_RAW_VAR] = f2u(t.raw_var[nb]);
        R[PR_NODE_VAR] = f2u(t.node_var[nb]);
        R[PR_NODE_VISITS] = static_cast<uint32_t>(t.node_visits[nb]);
#pragma unroll
        for (int c = 0; c < kPsA; ++c) {
            R[PR_LOGITS + c] = f2u(t.children_prior_logits[nb * kPsA + c]);
            M.gumbel[i][c] = t.root_gumbel[static_cast<long>(b) * kPsA + c];
            M.invalid[i][c] = t.root_invalid[static_cast<long>(b) * kPsA + c];
        }
        R[PR_STATE] = ps_pack(ds_load(t.embeddings + nb * sizeof(eaz_deepsea_state)));
        M.beta[i] = t.beta[b];
        M.D[i] = 0; M.parent[i] = 0; M.action[i] = 0; M.dst[i] = 0; M.hot[i] = 0;
    }

    const int r = lane & 15, kq = lane >> 4;  // MFMA operand coordinates of this lane
    const int K1 = a.net.in_dim;
    __syncthreads();

    // Both roles run the same barrier sequence per simulation:
    //   S1 hot[] + H1(head 0) | S2, S4, S6 layer-3 partial sums of head 0, 1, 2 (+ H1 of the next head) | S7 all head outputs
    if (helper) {
        // =========================================================== helper wave ===========================================================
        // layer-1 bias of the two heads whose W1 rows this wave gathers (lane = 4 columns)
        float hb1[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c) hb1[h][c] = a.net.b[hidv[h + 1] * 3][4 * lane + c];
        const float* W1g[2] = {a.net.w[hidv[1] * 3] + 4 * lane, a.net.w[hidv[2] * 3] + 4 * lane};
        __builtin_amdgcn_s_setprio(3);  // the youngest wave of the workgroup loses every issue arbitration against its SIMD partner otherwise
        const int n_fwd = (fused ? a.sp.T : 1) * (S + (fused ? 1 : 0));  // fused: every real step starts with the root's forward
        for (int f = 0; f < n_fwd && !ev_skip; ++f) {
            const int sim = fused ? f % (S + 1) - 1 : f;
            if (evaluate && sim == -1 && f > 0) {
                __syncthreads();  // S0 (evaluate): the real step before wrote its `terminated` flags
                if (all_terminated()) break;
            }
            EAZ_PSTAMP(0);
            __syncthreads();  // S1
            EAZ_PSTAMP(4);
            // W1 rows of the second head: in flight under layer 2 of the first (waves 0-3 issue nothing but their weight stream there)
            floatx4 gw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) gw[i] = *reinterpret_cast<const floatx4*>(W1g[0] + static_cast<long>(M.hot[4 * hw + i]) * kFcH);
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                if (h < 2) {
                    // H1 of head h + 1 into the other tile, under layer 2 of head h: relu(W1[hot] + b1) (exactly fmaf(1, w, +0) per column:
                    // every other term of the reference's chain is fmaf(0, w, acc) = acc)
                    float* H1n = H1 + ((h + 1) & 1) * kF16HFloats;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        floatx4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = __builtin_fmaf(1.0f, gw[i][c], 0.0f) + hb1[h][c];
                            o[i] = v > 0.0f ? v : 0.0f;
                        }
                        *reinterpret_cast<floatx4*>(H1n + h16_off(4 * lane + c, 4 * hw)) = o;
                    }
                }
                if (h == 0 && hw == 0 && lane < kPsTrees) {
                    // novelty lookup of the one-hot observation (hashes.py:23-38,162-229): the hash index of a row that is zero except
                    // obs[hot] = 1 depends on `hot` only and is tabulated once per network (NetDev::hot_index)
                    const uint32_t index = a.net.hot_index[M.hot[lane]];
                    const bool seen = a.net.bitset ? ((a.net.bitset[index >> 3] >> (index & 7u)) & 1u) != 0 : false;
                    M.o_nov[lane] = (seen ? 0.0f : 1.0f) * a.net.nov_scale;                                 // fully_connected.py:90
                }
                __syncthreads();  // S2 / S4 / S6: the partial sums of head h and H1 of head h + 1 are complete
                if (h == 0) {
                    // W1 rows of the third head (its H1 tile is the first head's, free from here on)
#pragma unroll
                    for (int i = 0; i < 4; ++i) gw[i] = *reinterpret_cast<const floatx4*>(W1g[1] + static_cast<long>(M.hot[4 * hw + i]) * kFcH);
                }
            }
            // ---- layer 3, second half + head activation, all three heads at once: lane = (row, output slot) with slot 0 = value, 1 = ube,
            // 2-3 = policy logits; ((c0 + c1) + (c2 + c3)) + b of the four quarter chains that waves 0-3 ran over their own hidden
            // columns (o_linear_parts) ----
            if (hw == 0) {
                const int row = lane & 15, slot = lane >> 4;
                const int hid = slot < 2 ? slot : a.head[2], o = slot < 2 ? 0 : slot - 2;
                const int hp = hidv[0] == hid ? 0 : (hidv[1] == hid ? 1 : 2);  // position of that head in this workgroup's order
                const float* pp = part + (hp * 4 * kPsTrees + row) * 4 + o;
                const float c0 = pp[0], c1 = pp[kPsTrees * 4], c2 = pp[2 * kPsTrees * 4], c3 = pp[3 * kPsTrees * 4];
                const float z = ((c0 + c1) + (c2 + c3)) + a.net.b[hid * 3 + 2][o];
                if (hid == 0) M.o_value[row] = det_tanh(z);                                     // fully_connected.py:55
                else if (hid == 1) M.o_ube[row] = (0.5f * (det_tanh(z) + 1.0f)) * a.net.max_u;  // :64, :93
                else M.o_logit[row][o] = z;                                                     // :68-81
            }
            EAZ_PSTAMP(5);
            __syncthreads();  // S7
        }
    } else {
        // ====================================================== waves 0-3: trees + layers 1-2 ======================================================
        // lane (r, kq) of wave w owns output columns n0 + 4r .. +3 (n0 = 64 w), k = 4 ks + kq
        const int n0 = wave * 64, col = n0 + 4 * r;
        float b2r[3][4];
        // W2 of the three heads as buffer resources: a refill is ONE buffer_load_dwordx4 with a per-lane byte offset that never changes
        // (row kq, columns col .. col+3) and the k-step in the scalar offset -- no per-refill address arithmetic on the vector ALU, and
        // half the address registers of a global_load (the wave issues one refill per four MFMAs and nothing else hides its issue cost)
        __amdgpu_buffer_rsrc_t W2r[3];
        const int voff = (kq * kFcH + col) * 4;
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            W2r[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.net.w[hidv[h] * 3 + 1]), 0, kFcH * kFcH * 4, 0x00020000);
#pragma unroll
            for (int c = 0; c < 4; ++c) b2r[h][c] = a.net.b[hidv[h] * 3 + 1][col + c];
        }
        const float* W1q0 = a.net.w[hidv[0] * 3] + 4 * lane;  // first head's row gather: lane = 4 columns, rows = this wave's trees
        const floatx4 b1q0 = *reinterpret_cast<const floatx4*>(a.net.b[hidv[0] * 3] + 4 * lane);
        // the W2 stream, kPsDepth k-steps ahead (k-step ks of a head reads rows 4 ks + kq); cyclic over the heads, so the first k-steps of
        // the next simulation are in flight during the tree phase
        floatx4 Bq[kPsDepth];
#pragma unroll
        for (int i = 0; i < kPsDepth; ++i) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(W2r[0], voff, i * 4096, 0));
        const EnvDev& env = a.env;
        // tree lanes: 16 per tree
        const int ti = tid >> 4, tj = tid & 15;
        uint32_t* T = trees + ti * L.tstride;
        uint16_t* P = path + ti * S;
        const int gb = tree0 + ti < B ? tree0 + ti : B - 1;
        const Group<1, kPsA> g = make_group<1, kPsA>(gb, 0, kPsA);
        const bool tree_live = tree0 + ti < B;  // trees beyond the batch mirror the last one; they never store outside LDS
        // fused self-play: beta * linspace(0, 1, B)[b] (selfplay.py:105, root_prep_kernel); search only: the caller's per-tree beta
        const float beta = evaluate ? a.sp.sbeta * 1.0f  // evaluate.py:34 beta * ones
                           : (fused ? a.sp.sbeta * (B > 1 ? static_cast<float>(gb) / static_cast<float>(B - 1) : 0.0f) : M.beta[ti]);
        const bool need_var = beta != 0.0f;
        auto amap_flip = [&](const eaz_deepsea_state& st) -> int {
            const int idx = ps_ds_row(st, env.G) * env.G + st.pos;                                         // deep_sea.py:62
            return (a.amap_lds ? amap[idx] : env.action_map[idx]) != 0;
        };

        const int n_steps = fused ? a.sp.T : 1;
        int steps_done = ev_skip ? 0 : n_steps;  // evaluate: this workgroup's trip count
        for (int step = 0; step < n_steps && !ev_skip; ++step) {
        Key kt{0u, 0u};
        if (fused) {
            // ---- a new real step: empty trees (mctx instantiate_tree_from_root, A.3: ints -1 / 0, floats 0), the root's observation ----
            kt = fold(a.sp.key, kTagStep + static_cast<uint32_t>(step));                                   // key_seq[t], selfplay.py:147
            for (int n = tj; n < N; n += 16) {
                uint32_t* R = T + n * PR_WORDS;
#pragma unroll
                for (int w = 0; w < PR_WORDS; ++w) R[w] = (w == PR_INDEX || w == PR_INDEX + 1) ? 0xFFFFFFFFu : 0u;
            }
            if (tj == 0) {
                if (step == 0) M.root_state[ti] = ps_pack(ds_load(a.sp.states + static_cast<long>(gb) * sizeof(eaz_deepsea_state)));
                const int hh = ds_hot_index(env, ps_unpack(M.root_state[ti]));                             // states.observation, selfplay.py:89-91
                M.hot[ti] = hh < 0 ? 0 : (hh >= K1 ? K1 - 1 : hh);
            }
            ps_wave_sync();
        }
        for (int sim = fused ? -1 : 0; sim <= S; ++sim) {
            EAZ_PSTAMP(0);
            if (sim >= 0) {
            const int D = M.D[ti], dst = M.dst[ti];
            if (sim > 0) {
                // ---- expand: mctx update_tree_node + topology (A.3) with epistemic_recurrent_fn (src/context.py:132-153) ------------
                const int parent = M.parent[ti], action = M.action[ti];
                if (tj == 0) {
                    float lg[kPsA] = {M.o_logit[ti][0], M.o_logit[ti][1]};
                    const float m = g.vmax(lg);                                                         // :135
#pragma unroll
                    for (int c = 0; c < kPsA; ++c) lg[c] = lg[c] - m;                                   // :137 (DeepSea: every action legal)
                    const bool term = M.o_term[ti] != 0;
                    const float value = term ? 0.0f : M.o_value[ti];                                    // :140
                    const float var = term ? 0.0f : ube_combine(M.o_ube[ti], M.o_nov[ti], t.max_u);     // :141
                    float disc = 1.0f * t.discount;                                                     // :142
                    if (t.two_player) disc = disc * -1.0f;                                              // :143
                    if (term) disc = 0.0f;                                                              // :144
                    uint32_t* Rn = T + dst * PR_WORDS;
                    Rn[PR_LOGITS] = f2u(lg[0]); Rn[PR_LOGITS + 1] = f2u(lg[1]);
                    Rn[PR_RAW_VALUE] = f2u(value); Rn[PR_NODE_VALUE] = f2u(value);
                    Rn[PR_RAW_VAR] = f2u(var); Rn[PR_NODE_VAR] = f2u(var);
                    Rn[PR_NODE_VISITS] += 1u;
                    uint32_t* Rp = T + parent * PR_WORDS;
                    Rp[PR_INDEX + action] = static_cast<uint32_t>(dst);
                    Rp[PR_REWARDS + action] = f2u(M.o_reward[ti]);                                      // :139
                    Rp[PR_DISCOUNTS + action] = f2u(disc);
                }
                ps_wave_sync();
                // ---- backward (A.6) ------------------------------------------------------------------------------------------
                const float leaf_v = u2f(T[dst * PR_WORDS + PR_RAW_VALUE]), leaf_u = u2f(T[dst * PR_WORDS + PR_RAW_VAR]);
                float Gin = leaf_v, Uin = leaf_u, cvin = leaf_v, cuin = leaf_u;
                for (int c = (D - 1) >> 4; c >= 0; --c) {
                    const int d = c * 16 + tj;
                    const bool valid = d < D;
                    const int e = P[valid ? d : 0];
                    const int pn = e & 0x7FFF, pa = e >> 15;
                    uint32_t* R = T + pn * PR_WORDS;
                    const float er = u2f(R[PR_REWARDS + pa]), ed = u2f(R[PR_DISCOUNTS + pa]);
                    const int cnt_i = static_cast<int>(R[PR_NODE_VISITS]), cvis = static_cast<int>(R[PR_VISITS + pa]);
                    const float nv = u2f(R[PR_NODE_VALUE]), nvar = u2f(R[PR_NODE_VAR]);
                    const int top = (D - c * 16 < 16 ? D - c * 16 : 16) - 1;  // deepest level of this chunk
                    // deepest level of the four trees of this wavefront: levels beyond it are skipped by a scalar branch
                    int wtop = __builtin_amdgcn_readlane(top, 0);
                    { const int t1 = __builtin_amdgcn_readlane(top, 16), t2 = __builtin_amdgcn_readlane(top, 32), t3 = __builtin_amdgcn_readlane(top, 48);
                      wtop = wtop > t1 ? wtop : t1; wtop = wtop > t2 ? wtop : t2; wtop = wtop > t3 ? wtop : t3; }
                    // every lane needs the edges of all deeper levels of its tree: through LDS (one 8-byte store, then wide broadcast loads)
                    // rather than 32 ds_bpermutes
                    *reinterpret_cast<float2*>(&M.edge[ti][tj][0]) = make_float2(er, ed);
                    ps_wave_sync();
                    float ers[16], eds[16];
#pragma unroll
                    for (int s4 = 0; s4 < 8; ++s4) {
                        const floatx4 v = *reinterpret_cast<const floatx4*>(&M.edge[ti][2 * s4][0]);
                        ers[2 * s4] = v[0]; eds[2 * s4] = v[1]; ers[2 * s4 + 1] = v[2]; eds[2 * s4 + 1] = v[3];
                    }
                    float G = Gin, U = Uin;
#pragma unroll
                    for (int s = 15; s >= 0; --s) {
                        if (s > wtop) continue;
                        const float rs = ers[s], dd = eds[s];
                        if (s <= top && s >= tj) {
                            G = rs + dd * G;
                            U = 0.0f + (dd * dd) * U;  // children_rewards_epistemic_variance == 0 (context.py:149)
                        }
                    }
                    const float cnt = static_cast<float>(cnt_i);
                    const float pv = (nv * cnt + G) / (cnt + 1.0f);
                    const float pvar = var_backup_update(t.var_backup, nvar, cnt, U);
                    float cv = __shfl_down(pv, 1, 16), cu = __shfl_down(pvar, 1, 16);  // node value of the level below
                    if (tj == top) { cv = cvin; cu = cuin; }
                    if (valid) {
                        R[PR_NODE_VALUE] = f2u(pv);
                        R[PR_NODE_VAR] = f2u(pvar);
                        R[PR_NODE_VISITS] = static_cast<uint32_t>(cnt_i + 1);
                        R[PR_VALUES + pa] = f2u(cv);
                        R[PR_VAR + pa] = f2u(cu);
                        R[PR_VISITS + pa] = static_cast<uint32_t>(cvis + 1);
                    }
                    Gin = __shfl(G, 0, 16); Uin = __shfl(U, 0, 16);
                    cvin = __shfl(pv, 0, 16); cuin = __shfl(pvar, 0, 16);
                }
                ps_wave_sync();
            }
            EAZ_PSTAMP(1);
            if (sim == S) break;
            // ---- re-score the nodes whose selection inputs changed: the path's levels 0 .. D-1 and the (re-)expanded node ---------
            const int nl = sim > 0 ? D + 1 : 1;
            for (int l = tj; l < nl; l += 16) {
                const int node = l < D ? (P[l] & 0x7FFF) : dst;
                uint32_t* R = T + node * PR_WORDS;
                NodeRows<kPsA> rw;
#pragma unroll
                for (int c = 0; c < kPsA; ++c) {
                    rw.logits[c] = u2f(R[PR_LOGITS + c]);
                    rw.values[c] = u2f(R[PR_VALUES + c]);
                    rw.rewards[c] = u2f(R[PR_REWARDS + c]);
                    rw.discounts[c] = u2f(R[PR_DISCOUNTS + c]);
                    rw.visits[c] = static_cast<int>(R[PR_VISITS + c]);
                    rw.index[c] = static_cast<int>(R[PR_INDEX + c]);
                    rw.var[c] = need_var ? u2f(R[PR_VAR + c]) : 0.0f;
                    rw.rvar[c] = 0.0f;
                }
                float cq[kPsA];
                qtransform<1, kPsA>(t, g, rw, u2f(R[PR_RAW_VALUE]), need_var ? u2f(R[PR_RAW_VAR]) : 0.0f, beta, cq);
                int act;
                if (node == 0) {
                    // gumbel_muzero_root_action_selection (A.4)
                    const int inv[kPsA] = {M.invalid[ti][0], M.invalid[ti][1]};
                    const float gum[kPsA] = {M.gumbel[ti][0], M.gumbel[ti][1]};
                    const int nvalid = (1 - (inv[0] != 0)) + (1 - (inv[1] != 0));
                    int simidx = rw.visits[0] + rw.visits[1];
                    const int nc = nvalid < t.max_considered ? nvalid : t.max_considered;
                    simidx = simidx > S - 1 ? S - 1 : simidx;
                    act = ps_root_pick(g, rw, cq, static_cast<float>(tab[nc * S + simidx]), gum, inv);
                } else {
                    // gumbel_muzero_interior_action_selection (A.4)
                    float x[kPsA], pr[kPsA];
#pragma unroll
                    for (int c = 0; c < kPsA; ++c) x[c] = rw.logits[c] + cq[c];
                    const int sumN = rw.visits[0] + rw.visits[1];
                    g.softmax(x, pr);
                    const float den = static_cast<float>(1 + sumN);
#pragma unroll
                    for (int c = 0; c < kPsA; ++c) x[c] = pr[c] - static_cast<float>(rw.visits[c]) / den;
                    act = g.argmax(x);
                }
                const int nx = act == 0 ? rw.index[0] : rw.index[1];
                R[PR_SEL] = (static_cast<uint32_t>(act) << 16) | (static_cast<uint32_t>(nx) & 0xFFFFu);
            }
            ps_wave_sync();
            EAZ_PSTAMP(2);
            // ---- select (A.4): walk the cached choices; DeepSea env.step of the chosen edge ---------------------------------------
            if (tj == 0) {
                int node = 0, depth = 0, act, nx;
                while (true) {
                    const uint32_t sw = T[node * PR_WORDS + PR_SEL];
                    act = static_cast<int>(sw >> 16);
                    nx = static_cast<int>(static_cast<int16_t>(sw & 0xFFFFu));
                    P[depth] = static_cast<uint16_t>(node | (act << 15));
                    depth += 1;
                    if (nx < 0 || depth >= t.max_depth) break;
                    node = nx;
                }
                const int nd = nx < 0 ? sim + 1 : nx;
                M.parent[ti] = node; M.action[ti] = act; M.dst[ti] = nd; M.D[ti] = depth;
                const eaz_deepsea_state ps0 = ps_unpack(T[node * PR_WORDS + PR_STATE]);
                const eaz_deepsea_state ns = ps_ds_step(ps0, act, amap_flip(ps0), env.G);  // context.py:125-127
                T[nd * PR_WORDS + PR_STATE] = ps_pack(ns);
                M.o_reward[ti] = ns.reward;   // rewards[b, current_player], single player (deep_sea.py:100-102)
                M.o_term[ti] = ns.terminated;
                const int hh = ds_hot_index(env, ns);
                M.hot[ti] = hh < 0 ? 0 : (hh >= K1 ? K1 - 1 : hh);
            }
            }  // sim >= 0
            EAZ_PSTAMP(3);
            ps_wave_sync();  // hot[] of this wave's four trees
            // -------------------------------------------------------- forward of the 16 leaves --------------------------------------------
            {
                // layer 1 of the first head: relu(W1[hot] + b1) for the rows of THIS wave's trees (4 wave .. 4 wave + 3), lane = 4 columns: no
                // workgroup barrier between the select and the gather, four 16-byte loads per lane
                floatx4 w1n[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w1n[k] = *reinterpret_cast<const floatx4*>(W1q0 + static_cast<long>(M.hot[4 * wave + k]) * kFcH);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    floatx4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float v = __builtin_fmaf(1.0f, w1n[k][c], 0.0f) + b1q0[c];
                        o[k] = v > 0.0f ? v : 0.0f;
                    }
                    *reinterpret_cast<floatx4*>(H1 + h16_off(4 * lane + c, 4 * wave)) = o;
                }
            }
            __syncthreads();  // S1: H1 of head 0; hot[], the leaves' rewards / terminated flags
            EAZ_PSTAMP(4);
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                // ---- layer 2: [16 x 256] x [256 x 256]; the weight stream runs kPsDepth k-steps ahead and on into the next head (next
                // simulation); A operands come from LDS kPsARing k-steps ahead.  The first round of heads 1 and 2 also carries the
                // layer-3 quarter chain of the head before: 16 dependent MFMAs, one per k-step, hidden between independent ones. ----
                floatx4 acc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                EAZ_PSTAMP(6 + 3 * h);
                const float* Ap = H1 + (h & 1) * kF16HFloats + kq * 16 + r;  // h16_off(4 ks + kq, r) = 68 ks + 16 kq + r
                float av[kPsARing];
#pragma unroll
                for (int i = 0; i < kPsARing; ++i) av[i] = Ap[i * 68];
                // quarter chain of head h - 1: A = the H2 columns this wave wrote before the barrier, B = W3 rows of its quarter of k
                floatx4 a3 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                const float* hpA = H2 + h16_off(n0 + kq, r);                                       // H2[h16_off(n0 + 4 ks + kq, r)] = hpA[68 ks]
                const float* wpB = W3s + ((h > 0 ? h - 1 : 0) * 4 + (r & 3)) * kFcH + n0 + kq;     // W3[n0 + 4 ks + kq][r] (columns >= 4 repeat 0..3, dropped)
                // first round (peeled: it carries the chain), then the other rounds as a real loop -- with all 192 refill addresses of the
                // three heads loop-invariant the compiler would precompute and spill them
                if (h > 0) ps_round<true>(acc, Bq, av, Ap + kPsARing * 68, W2r[h], voff, 4096 * kPsDepth, a3, hpA, wpB);
                else ps_round<false>(acc, Bq, av, Ap + kPsARing * 68, W2r[h], voff, 4096 * kPsDepth, a3, hpA, wpB);
                if (h > 0 && r < 4) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) part[(((h - 1) * 4 + wave) * kPsTrees + 4 * kq + reg) * 4 + r] = a3[reg];
                }
#pragma unroll 1
                for (int rd = 1; rd < 64 / kPsDepth; ++rd) {
                    const bool same = rd + 1 < 64 / kPsDepth;  // the refills of the last round belong to the next head
                    const __amdgpu_buffer_rsrc_t rs = same ? W2r[h] : W2r[h == 2 ? 0 : h + 1];
                    const int sbase = same ? 4096 * kPsDepth * (rd + 1) : 0;
                    // reads past k-step 63 land in the next tile and are never used
                    ps_round<false>(acc, Bq, av, Ap + (rd * kPsDepth + kPsARing) * 68, rs, voff, sbase, a3, hpA, wpB);
                }
                // relu(acc + b2) -> H2: lane (r, kq) holds rows 4 kq .. +3 of column n0 + 4r + c.  Every wave reads back only the columns it
                // wrote (the transposition of its D tile into A operands of the quarter chain), so the tile needs no workgroup barrier of its
                // own: DS operations of a wave execute in order, and the next head's stores come 8 k cycles after the last read-back.
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    floatx4 o;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float v = acc[c][reg] + b2r[h][c];
                        o[reg] = v > 0.0f ? v : 0.0f;
                    }
                    *reinterpret_cast<floatx4*>(H2 + h16_off(col + c, 4 * kq)) = o;
                }
                if (h == 2) {
                    // ---- the last head's quarter chain has no layer 2 to hide under: 16 dependent MFMAs (k ascending = the oracle's fmaf chain
                    // of that quarter) ----
                    const float* hpA = H2 + h16_off(n0 + kq, r);
                    const float* wpB = W3s + (h * 4 + (r & 3)) * kFcH + n0 + kq;
                    float wb[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) wb[q] = wpB[4 * q];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    floatx4 z3 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int q = 0; q < 16; ++q) z3 = __builtin_amdgcn_mfma_f32_16x16x4f32(hpA[68 * q], wb[q], z3, 0, 0, 0);
                    if (r < 4) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) part[((h * 4 + wave) * kPsTrees + 4 * kq + reg) * 4 + r] = z3[reg];
                    }
                }
                EAZ_PSTAMP(7 + 3 * h);
                __syncthreads();  // S2 / S4 / S6: H1 of head h + 1 (helper) visible; after S6 all partial sums are
            }
            EAZ_PSTAMP(5);
            __syncthreads();  // S7: the heads' outputs
            if (sim < 0) {
                // ---- fused self-play, root: selfplay.py:92-106,113 (root_prep_kernel) + mctx instantiate_tree_from_root (tree_root_kernel) ----
                if (tj == 0) {
                    const float root_v = M.o_value[ti];
                    const float root_u = ube_combine(M.o_ube[ti], M.o_nov[ti], t.max_u);                  // fully_connected.py:92-96
                    const Key gkey = evaluate ? Key{0u, 0u} : fold(kt, kTagGumbel);  // evaluate: gumbel_scale = 0 (evaluate.py:44), the host driver's key
                    float lg[kPsA], gum[kPsA];
#pragma unroll
                    for (int c = 0; c < kPsA; ++c) {
                        lg[c] = (a.sp.flags & kSpUniform) ? 1.0f : M.o_logit[ti][c];                      // :96-98
                        const long gi = static_cast<long>(gb) * kPsA + c;
                        const float gn = a.sp.gumbel ? a.sp.gumbel[static_cast<long>(step) * B * kPsA + gi] : gumbel(gkey, static_cast<uint32_t>(gi));
                        gum[c] = a.sp.gumbel_scale * gn;
                        M.gumbel[ti][c] = gum[c];
                        M.invalid[ti][c] = 0;                                                             // :113 ~legal_action_mask: all legal (deep_sea.py:19)
                    }
                    const float m = g.vmax(lg);                                                           // mask_invalid_actions (A.2)
                    T[PR_LOGITS] = f2u(lg[0] - m); T[PR_LOGITS + 1] = f2u(lg[1] - m);
                    T[PR_RAW_VALUE] = f2u(root_v); T[PR_NODE_VALUE] = f2u(root_v);
                    T[PR_RAW_VAR] = f2u(root_u); T[PR_NODE_VAR] = f2u(root_u);
                    T[PR_NODE_VISITS] = 1u;
                    T[PR_STATE] = M.root_state[ti];
                    M.root_v[ti] = root_v; M.root_u[ti] = root_u;
                    M.D[ti] = 0; M.parent[ti] = 0; M.action[ti] = 0; M.dst[ti] = 0;
                }
                ps_wave_sync();
            }
        }
        if (fused && tj == 0) {
            // ---- fused self-play, after the search: gumbel_muzero_policy tail (A.2, tree_finish_kernel), epistemic_summary (A.7,
            // tree_summary_kernel), action choice (selfplay.py:118-134, choose_action_kernel), auto-reset env.step (:135), SelfplayOutput[t] ----
            NodeRows<kPsA> rw;
#pragma unroll
            for (int c = 0; c < kPsA; ++c) {
                rw.logits[c] = u2f(T[PR_LOGITS + c]); rw.values[c] = u2f(T[PR_VALUES + c]); rw.rewards[c] = u2f(T[PR_REWARDS + c]);
                rw.discounts[c] = u2f(T[PR_DISCOUNTS + c]); rw.visits[c] = static_cast<int>(T[PR_VISITS + c]);
                rw.index[c] = static_cast<int>(T[PR_INDEX + c]); rw.var[c] = need_var ? u2f(T[PR_VAR + c]) : 0.0f; rw.rvar[c] = 0.0f;
            }
            float cq[kPsA];
            qtransform<1, kPsA>(t, g, rw, u2f(T[PR_RAW_VALUE]), u2f(T[PR_RAW_VAR]), beta, cq);
            const int inv[kPsA] = {M.invalid[ti][0], M.invalid[ti][1]};
            const float gum[kPsA] = {M.gumbel[ti][0], M.gumbel[ti][1]};
            const int maxN = rw.visits[0] > rw.visits[1] ? rw.visits[0] : rw.visits[1];
            int act = ps_root_pick(g, rw, cq, static_cast<float>(maxN), gum, inv);
            float x[kPsA], aw[kPsA];
#pragma unroll
            for (int c = 0; c < kPsA; ++c) x[c] = rw.logits[c] + cq[c];
            const float mx = g.vmax(x);
#pragma unroll
            for (int c = 0; c < kPsA; ++c) x[c] = inv[c] ? kF32Min : x[c] - mx;
            g.softmax(x, aw);
            // summary
            const float sum_value = u2f(T[PR_NODE_VALUE]), sum_std = __builtin_sqrtf(u2f(T[PR_NODE_VAR]));
            const float ftot = static_cast<float>(rw.visits[0] + rw.visits[1]);
            float qvar[kPsA], vprob[kPsA];
#pragma unroll
            for (int c = 0; c < kPsA; ++c) {
                const float d = rw.discounts[c];
                qvar[c] = 0.0f + (d * d) * u2f(T[PR_VAR + c]);  // children_rewards_epistemic_variance == 0
                vprob[c] = static_cast<float>(rw.visits[c]) / (ftot > 1.0f ? ftot : 1.0f);
            }
            // action choice: categorical over visit PROBABILITIES / action weights used as logits (reproduced literally), = argmax(x + Gumbel)
            if (a.sp.flags & kSpSampleVisits) {
                const Key kvis = fold(kt, kTagSampleVisits);
                float best = -kInf; int bi = 0;
#pragma unroll
                for (int c = 0; c < kPsA; ++c) {
                    const float v = vprob[c] + gumbel(kvis, static_cast<uint32_t>(gb * kPsA + c));
                    if (v > best) { best = v; bi = c; }
                }
                act = bi;
            }
            if (a.sp.flags & kSpSampleImproved) {
                const Key kimp = fold(kt, kTagSampleImproved);
                float best = -kInf; int bi = 0;
#pragma unroll
                for (int c = 0; c < kPsA; ++c) {
                    const float v = aw[c] + gumbel(kimp, static_cast<uint32_t>(gb * kPsA + c));
                    if (v > best) { best = v; bi = c; }
                }
                act = bi;
            }
            // vmap(auto_reset(env.step, env.init)) (selfplay.py:26-75,135): a finished episode restarts, anything else steps
            const eaz_deepsea_state cur = ps_unpack(M.root_state[ti]);
            // evaluate: a plain env.step (evaluate.py:47; a finished episode takes no-op steps with zero rewards), rewards summed (:48-50)
            const eaz_deepsea_state nxt = (!evaluate && (cur.terminated | cur.truncated)) ? ds_init() : ps_ds_step(cur, act, amap_flip(cur), env.G);
            M.root_state[ti] = ps_pack(nxt);
            if (evaluate) {
                M.sum_reward[ti] = M.sum_reward[ti] + nxt.reward;
                if (tree_live) M.real_term[ti] = nxt.terminated != 0;
            }
            if (tree_live) {
                const long o = static_cast<long>(step) * B + gb;
                if (a.sp.out_value_pred) a.sp.out_value_pred[o] = M.root_v[ti];
                if (a.sp.out_ube_pred) a.sp.out_ube_pred[o] = M.root_u[ti];
                if (a.sp.out_root_value) a.sp.out_root_value[o] = sum_value;
                if (a.sp.out_root_std) a.sp.out_root_std[o] = sum_std;
                if (a.sp.out_action) a.sp.out_action[o] = act;
                if (a.sp.out_qvar) { a.sp.out_qvar[o * kPsA] = qvar[0]; a.sp.out_qvar[o * kPsA + 1] = qvar[1]; }
                if (a.sp.out_states) ds_store(a.sp.out_states + o * sizeof(eaz_deepsea_state), nxt);
                if (step == n_steps - 1 || evaluate) ds_store(a.sp.states + static_cast<long>(gb) * sizeof(eaz_deepsea_state), nxt);
            }
        }
        if (fused) ps_wave_sync();
        if (evaluate && step + 1 < n_steps) {
            __syncthreads();  // S0 (the helper waves hold the same barrier in front of the next root forward)
            if (all_terminated()) { steps_done = step + 1; break; }  // cond_fn, evaluate.py:18-20
        }
        }  // real steps
        if (evaluate) {
            if (tj == 0 && tree_live) a.sp.eval_sum[gb] = M.sum_reward[ti];
            if (tid == 0) { a.sp.eval_wg_iters[blockIdx.x] = steps_done; atomicMax(a.sp.eval_iters, steps_done); }
        }
    }

    if (fused) return;  // self-play never reads the tree arrays
    // ---------------------------------------------------------------- write the trees back (mctx SoA layout) -----------------------
    __syncthreads();
    int16_t* par = reinterpret_cast<int16_t*>(H1);            // [16][N] parent, then [16][N] action_from_parent (H1/H2 are free now)
    int16_t* afp = par + kPsTrees * N;
    for (int i = tid; i < 2 * kPsTrees * N; i += kPsThreads) par[i] = -1;
    __syncthreads();
    for (int i = tid; i < kPsTrees * N * kPsA; i += kPsThreads) {
        const int tr = i / (N * kPsA), rem = i - tr * (N * kPsA), n = rem / kPsA, c = rem - n * kPsA;
        const int ch = static_cast<int>(trees[tr * L.tstride + n * PR_WORDS + PR_INDEX + c]);
        if (ch >= 0) { par[tr * N + ch] = static_cast<int16_t>(n); afp[tr * N + ch] = static_cast<int16_t>(c); }
    }
    __syncthreads();
    const int ntree = B - tree0 < kPsTrees ? B - tree0 : kPsTrees;
    const long nb0 = static_cast<long>(tree0) * N;
    for (int i = tid; i < ntree * N; i += kPsThreads) {
        const int tr = i / N, n = i - tr * N;
        const uint32_t* R = trees + tr * L.tstride + n * PR_WORDS;
        t.raw_values[nb0 + i] = u2f(R[PR_RAW_VALUE]);
        t.node_values[nb0 + i] = u2f(R[PR_NODE_VALUE]);
        t.raw_var[nb0 + i] = u2f(R[PR_RAW_VAR]);
        t.node_var[nb0 + i] = u2f(R[PR_NODE_VAR]);
        t.node_visits[nb0 + i] = static_cast<int32_t>(R[PR_NODE_VISITS]);
        t.parents[nb0 + i] = par[i];
        t.action_from_parent[nb0 + i] = afp[i];
        if (n > 0 && R[PR_NODE_VISITS] != 0u) ds_store(t.embeddings + (nb0 + i) * sizeof(eaz_deepsea_state), ps_unpack(R[PR_STATE]));
    }
    for (int i = tid; i < ntree * N * kPsA; i += kPsThreads) {
        const int tr = i / (N * kPsA), rem = i - tr * (N * kPsA), n = rem / kPsA, c = rem - n * kPsA;
        const uint32_t* R = trees + tr * L.tstride + n * PR_WORDS;
        const long o = nb0 * kPsA + i;
        t.children_index[o] = static_cast<int32_t>(R[PR_INDEX + c]);
        t.children_visits[o] = static_cast<int32_t>(R[PR_VISITS + c]);
        t.children_prior_logits[o] = u2f(R[PR_LOGITS + c]);
        t.children_values[o] = u2f(R[PR_VALUES + c]);
        t.children_rewards[o] = u2f(R[PR_REWARDS + c]);
        t.children_discounts[o] = u2f(R[PR_DISCOUNTS + c]);
        t.children_var[o] = u2f(R[PR_VAR + c]);
        t.children_reward_var[o] = 0.0f;
    }
}

}  // namespace eaz
