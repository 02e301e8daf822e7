// search_subleq_persist.hpp -- subleq16_selfplay_kernel: a whole selfplay() call for Subleq word_size 16 in ONE launch (C3 production path);
// with PersistSelfplay::on == 0 the same kernel is ONE search (eaz_search_run: the emctx call shape, reanalyze, evaluate): root from, whole
// tree back to the mctx structure-of-arrays layout of tree.hpp.
//
// Reference: step_fn of src/selfplay.py:86-143 (root inference :89-91, root preparation :92-106, emctx policy :107-117 [EXT] = SURVEY.md
// A.2-A.7, action choice :118-134, auto-reset env.step :135, SelfplayOutput :136-143), with the recurrent fn of src/context.py:117-155
// (env.step of src/envs/subleq.py:648-677 + forward of src/network/fully_connected.py:41-101 + XXHash novelty of hashes.py:162-229).
//
// One workgroup of 512 threads (8 waves) owns 16 environments / trees for all T real steps.  Nothing of a tree lives in global memory
// except the prior logits of its nodes and their softmax (a [N,32] f32 scratch row set per tree: both are functions of the node alone,
// written once at expansion; read back only for the nodes of the backed-up path, requested when the select ends, i.e. a whole
// interpreter + forward phase before the re-score uses them):
//   trees   LDS node records (QN_WORDS words).  A tree of N nodes has at most N - 1 edges whatever A is, so the [N,A] child arrays of
//           mctx collapse into per-NODE fields: children_values / _visits / _variance[p, a] are always the node value / visit count /
//           variance of the child itself (the backward pass copies them on every visit, A.6), the edge's reward and "discount is zero"
//           flag sit in the child's record, and children_index is 16 bytes per node.
//   states  LDS, 32 bytes per node: the sixteen 4-bit words of the program as one 64-bit value, input_after / output_after as 8 bytes
//           each, a header word; example_input / example_output are constants of the task (subleq.py:398-501).
// Per simulation (lanes 32 ti .. 32 ti + 31 belong to tree ti, i.e. a tree never leaves its wavefront during the tree phase):
//   T  expand + backward of the previous leaf (lane = level of the recorded path, as search_persist.hpp), re-score of the D + 1 nodes whose
//      selection inputs changed (4 lanes x 4 actions per node: Group<4,4>, the reductions are two DPP quad permutes), select = a walk
//      through the cached choices;
//   E  env.step of the 16 chosen edges by the all-scalar interpreter of env_kernels.hpp (subleq16_wave_run), one environment per
//      wavefront at a time from a work queue, LDS to LDS;
//   F  the forward of the three heads self-play reads for the 16 new observations, heads in turn on waves 0-3 (wave w owns 64 hidden
//      columns) with the weight stream of search_persist.hpp: ONE continuous buffer-load stream W1(h) | W2(h) | W1(h+1) ... 16 k-steps
//      ahead in registers, cyclic over the heads, so the first k-steps of the next simulation are in flight during T and E.  The
//      observation never becomes an f32 tile: a Subleq observation is 240 bits, the interpreter's wave leaves them as 256 FP8 bytes per
//      row (0x38 = 1.0) in the operand order of the MFMA lanes (QpMisc::obytes), a lane reads the 16 k-steps of a round as one 16-byte
//      word and one v_cvt_pk_f32_fp8 makes the A operands of two k-steps.  Layer 3 = quarter chains whose operands W3 stay in registers
//      for the whole launch; a head's chain rides in a round of the next head (ps_round).  The heads' outputs are never
//      materialised: expand / root preparation read the four quarter sums and apply ((c0 + c1) + (c2 + c3)) + b themselves.
// Same float operations in the same order as tree_step_kernel<16,1> / fc16p_forward_kernel (the 16-slot butterfly sum is
// lsum<4> + two quad exchanges), so results are bit-identical to the per-simulation path and to the oracle (tests/conftest.py variants).
#pragma once
#include "net_fc_pipe.hpp"
#include "search_kernels.hpp"

namespace eaz {

constexpr int kQpTrees = 16;
constexpr int kQpThreads = 512;
constexpr int kQpA = 16;

// node record in LDS (32-bit words)
enum : int {
    QN_RAW_VALUE = 0, QN_NODE_VALUE = 1, QN_RAW_VAR = 2, QN_NODE_VAR = 3,
    QN_REWARD = 4,  // children_rewards[parent, action_from_parent]
    QN_META = 5,    // node_visits (bits 0-7) | parent (8-15) | action_from_parent (16-19) | edge discount is zero (20)
    QN_SEL = 6,     // cached selection: action << 8 | child (0xFF: unvisited)
    QN_CHILD = 7,   // children_index as 16 bytes (0xFF: unvisited), action a in byte a & 3 of word a >> 2
    QN_WORDS = 11
};
// compact Subleq-16 state (32-bit words): the program as 16 nibbles, input_after / output_after as bytes (W = 16 beyond the valid ones)
enum : int { QS_MEM = 0, QS_INAF = 2, QS_OUTAF = 4, QS_HDR = 6, QS_REWARD = 7, QS_WORDS = 8 };
// header: step_count (bits 0-7) | terminated 8 | truncated 9 | solved 10 | rows-from-root 11 | task (12..31, signed)
// rows-from-root: example_input / example_output of this state are the root blob's (never recomputed since); otherwise the task's table

struct QpMisc {
    int D[kQpTrees], parent[kQpTrees], action[kQpTrees], dst[kQpTrees];
    float o_nov[kQpTrees], o_reward[kQpTrees];
    int o_term[kQpTrees];
    float gumbel[kQpTrees][kQpA];
    float root_logits[kQpTrees][kQpA];  // masked prior logits of node 0
    float root_probs[kQpTrees][kQpA];   // their softmax (qtransform's prior probabilities of the node)
    float new_logits[kQpTrees][kQpA];   // masked prior logits of the node expanded by this simulation
    float new_probs[kQpTrees][kQpA];
    float edge[kQpTrees][16][2];        // backward: (reward, discount) of the 16 path levels of the chunk being folded
    float root_v[kQpTrees], root_u[kQpTrees];
    uint32_t root_state[kQpTrees][QS_WORDS];
    uint32_t root_ex[kQpTrees][4];      // example_input / example_output rows of the root blob (bytes)
    int real_action[kQpTrees];
    int env_next;
    int pad[3];
    alignas(16) uint8_t qkey[kQpTrees]; // expected cost of the environment's next interpreter step (queue order)
    int16_t task[kQpTrees];             // the environments' tasks: constant between two resets (subleq.py:624), so not stored per node
    float beta[kQpTrees];               // selfplay.py:105 beta * linspace(0, 1, B)[b]; search only: the caller's per-tree beta
    int nc[kQpTrees];                   // min(valid root actions, max_num_considered_actions): row of the considered-visit table
    uint8_t invalid[kQpTrees][kQpA];    // root_invalid_actions (self-play: none, subleq.py:626)
    float o_value[kQpTrees], o_ube[kQpTrees];
    float b3[20];                       // output biases: value, ube, the 16 policy logits
    float b12[3][2][kFcH];              // hidden-layer biases of the three heads (layer 1, layer 2)
    alignas(16) uint8_t obytes[kQpTrees][256];  // the 16 observations as layer-1 A operands: [row][64 kq + ks] = input 4 ks + kq of the row as an FP8 (OCP e4m3)
                                                // byte, 0x38 = 1.0 or 0 (inputs 240..255: 0) -- lane (row, kq) reads its 64 k-steps as four 16-byte words and
                                                // one v_cvt_pk_f32_fp8 makes the operands of TWO k-steps (subleq.py:88-98)
    uint8_t words[kQpTrees][48];        // row-words of the 16 observations in flight, 5 observation bits each (subleq.py:88-98)
    float sum_reward[kQpTrees];         // evaluate mode: sum of the real steps' rewards (evaluate.py:48-50)
    uint8_t real_term[kQpTrees];        // evaluate mode: `terminated` of the environments' current states
    uint8_t l1var[kQpTrees];            // layer-1 mode 1: which start vector the observation in flight chains from (0: the task's example rows, 1: the root blob's)
};
constexpr int kQpPartLd = 20;  // quarter sums [4 waves][16 rows][kQpPartLd]: slot 0 value, 1 ube, 2..17 policy logits

struct QpLds {
    int tstride;  // words per tree
    int off_h1, off_h2, off_part, off_tree, off_state, off_path, off_misc, total;
};
__host__ __device__ inline QpLds qp_lds(int N, int S) {
    QpLds L;
    int o = 0;
    L.off_h1 = o; o += 2 * kF16HBytes;  // two H1 tiles: head h + 1 writes its tile while slower waves may still read head h's
    L.off_h2 = o; o += kF16HBytes;      // every wave reads back only the 64 columns it wrote
    L.off_part = o; o += 4 * 16 * kQpPartLd * 4;
    L.tstride = (N * QN_WORDS) | 1;
    L.off_tree = o; o += kQpTrees * L.tstride * 4;
    o = (o + 15) & ~15;
    L.off_state = o; o += kQpTrees * N * QS_WORDS * 4;
    L.off_path = o; o += (kQpTrees * S * 2 + 15) & ~15;
    L.off_misc = o; o += (static_cast<int>(sizeof(QpMisc)) + 15) & ~15;
    L.total = o;
    return L;
}

struct Sq16SelfplayArgs {
    PersistSelfplay sp;  // search_persist.hpp: the call's T, key, flags, beta, state / noise / output pointers
    TreeDev t;           // options, table; children_prior_logits is NOT used (see `logits`)
    EnvDev e;
    NetDev net;
    HeadList heads;      // value, ube, the policy head the search reads (context.py:132)
    float* logits;       // [ceil16(B), N, 32] scratch rows of the trees: masked prior logits [16] | their softmax [16]
    float* l1init;       // layer-1 mode 1, start of the layer-1 chains (qp_l1_start): [6 task branches][3 heads][256] for the example rows of a task's
                         // table (written by sq16_l1_table_kernel in front of every launch), then [ceil16(B)][3][256] for the example rows of the trees'
                         // root blobs (written by the kernel's prologue; read only for states that carry the blob's rows)
    unsigned long long* stamps;
};

#ifdef EAZ_FC_STAMPS
// diagnostic build: shader-clock stamps of simulation S / 2 of the first real step, lane 0 of every wave: [256 workgroups][8 waves][16]
// (slots 8..15: inside the first interpreter step of the wave, EAZ_QESTAMP)
#define EAZ_QPSTAMP(i)                                                                                 \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (step == 0 && s == 1 + S / 2 && (tid & 63) == 0 && blockIdx.x < 256 && qp_args()->stamps) { \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            qp_args()->stamps[(blockIdx.x * 8 + (tid >> 6)) * 16 + (i)] = t_;                          \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
// raw stamp `i` of the wave's FIRST interpreter step of the stamped simulation (sp = the wave's stamp row or null)
#define EAZ_QESTAMP(sp, i)                                                                             \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if ((sp) != nullptr && lane == 0) {                                                            \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            (sp)[(i)] = t_;                                                                            \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
__device__ unsigned long long g_qp_fine[2048 * 8];  // diagnostic build: raw stamps inside the interpreter segment of the stamped step, [workgroup * 8 + wave][8]
#define EAZ_QFSTAMP(sp, i)                                                                             \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if ((sp) != nullptr && lane == 0 && blockIdx.x < 256) {                                        \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            g_qp_fine[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = t_;                           \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
__device__ unsigned long long g_qp_fwd[2048 * 8];  // diagnostic build: stamps inside the forward of the stamped simulation, [workgroup * 8 + wave][8]: start, then
                                                   // after layer 1 / layer 2 of head 0, 1, 2, end
#define EAZ_QWSTAMP(i)                                                                                 \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (step == 0 && s == 1 + S / 2 && (tid & 63) == 0 && blockIdx.x < 256) {                      \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            g_qp_fwd[(blockIdx.x * 8 + (tid >> 6)) * 8 + (i)] = t_;                                    \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
__device__ unsigned long long g_qp_rounds[2048 * 64];  // diagnostic build: stamps after every half-round of 8 k-steps of the stamped forward, [workgroup * 8 + wave][head * 16 + layer * 8 + half-round]
#define EAZ_QRSTAMP(i)                                                                                 \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (step == 0 && s == 1 + S / 2 && (tid & 63) == 0 && blockIdx.x < 256) {                      \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            g_qp_rounds[(blockIdx.x * 8 + (tid >> 6)) * 64 + (i)] = t_;                                \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
__device__ uint32_t g_qp_its[64 * 64 * 16];  // diagnostic build, first real step of workgroups 0..63: [workgroup][simulation][environment] = cost of the step | parent's cost << 8 | written << 16
__device__ unsigned long long g_qp_ehist[128];  // diagnostic build: interpreter steps of workgroups 0..63 -- [0..3] by test cases run, [4..35] by interpreted cycles / 4 of
                                                 // the whole step, [36..67] of test case 0 alone, [68..99] of the steps that ran more than one case; [100] sum of cycles
__device__ unsigned long long g_qp_dhist[72];  // diagnostic build: histogram of the select depth D over all trees and simulations (+ [64..71]: by simulation octile, sum of D)
#else
#define EAZ_QPSTAMP(i)
#define EAZ_QESTAMP(sp, i)
#define EAZ_QFSTAMP(sp, i)
#define EAZ_QWSTAMP(i)
#define EAZ_QRSTAMP(i)
#endif

__device__ __forceinline__ unsigned long long qp_u64(uint32_t lo, uint32_t hi) { return static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32); }
// sixteen bytes (one program word each) -> sixteen nibbles
__device__ __forceinline__ unsigned long long qp_nibbles(const uint4& c) {
    const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
    unsigned long long m = 0ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = cw[i];
        const uint32_t n4 = (w & 0xFu) | ((w >> 4) & 0xF0u) | ((w >> 8) & 0xF00u) | ((w >> 12) & 0xF000u);
        m |= static_cast<unsigned long long>(n4) << (16 * i);
    }
    return m;
}
// eight uint16 values (two uint4) -> eight bytes
__device__ __forceinline__ unsigned long long qp_row_bytes(const uint4& r) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    unsigned long long v = 0ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) v |= static_cast<unsigned long long>((w[i] & 0xFFu) | ((w[i] >> 8) & 0xFF00u)) << (16 * i);
    return v;
}
// header word of a compact state: step count (8 bits) | terminated, truncated, solved, "example rows from the root blob" | cost hint (4 bits) |
// touched mask (16 bits).  The last two are not part of the environment state (they never leave LDS): `touched` bit l = the interpreter runs
// whose results this state carries (input_after, output_after, solved) fetched, read or wrote program word l -- all ones when unknown (a
// caller's root blob, states no run produced); `hint` = min(15, interpreted cycles / 4) of those runs.  The task is a constant of the tree
// between two resets (QpMisc::task).
__device__ __forceinline__ int qp_hdr_hint(uint32_t hdr) { return static_cast<int>((hdr >> 12) & 15u); }
__device__ __forceinline__ uint32_t qp_hdr_touched(uint32_t hdr) { return hdr >> 16; }
__device__ __forceinline__ uint32_t qp_hdr(int step, int term, int trunc, int solved, int exroot, int hint = 0, uint32_t touched = 0xFFFFu) {
    return static_cast<uint32_t>(step & 0xFF) | (static_cast<uint32_t>(term != 0) << 8) | (static_cast<uint32_t>(trunc != 0) << 9) |
           (static_cast<uint32_t>(solved != 0) << 10) | (static_cast<uint32_t>(exroot != 0) << 11) | (static_cast<uint32_t>(hint & 15) << 12) |
           (touched << 16);
}
// observation code of a row-word: bits 0..3 = v mod 16, bit 4 = (v == 16)  (subleq.py:88-98)
__device__ __forceinline__ uint8_t qp_code(uint32_t v) { return static_cast<uint8_t>((v & 15u) | (v == 16u ? 16u : 0u)); }

// Layer 1's packed input order (Sq16 one-launch kernel only; sq16_pack_w1_kernel builds the device copy of W1 in the same order).  An input that is
// structurally zero contributes fmaf(0, w, acc) = acc, bit for bit, so it is left out of the chain; the others keep their ascending order:
//   mode 0: 224 inputs -- the four value bits of the 16 memory words (a memory word is < 16 here, so its pad flag, subleq.py:88-98, is always
//           0: 16 of the 240 inputs never fire), then the four IO rows as they are (observation inputs 80..239);
//   mode 1: 144 inputs -- memory bits, then input_after and output_after; the example_input / example_output rows are constants of the tree
//           and start the chain as a per-tree partial sum (the network's "constant inputs first" input order, NetDev::order).
// Returns the observation index of packed input kp, or -1 for a zero row of the device copy.
__host__ __device__ inline int sq16_l1_src(int mode, int kp) {
    if (kp < 64) return 5 * (kp >> 2) + (kp & 3);
    const int q = kp - 64;
    if (mode == 0) return q < 160 ? 80 + q : -1;
    return q < 40 ? 120 + q : (q < 80 ? 160 + q : -1);
}
__host__ __device__ inline int sq16_l1_halves(int mode) { return mode == 0 ? 7 : 5; }  // half-rounds of 8 k-steps (32 inputs) of layer 1
// device copy of a head's layer-1 weights in the packed order: out[kp][c] = W1[sq16_l1_src(kp)][c], zero rows where the source is -1;
// grid = (32 x half-rounds, 4 heads), 256 threads = the hidden columns
__global__ void __launch_bounds__(256) sq16_pack_w1_kernel(NetDev net, int mode, float* o0, float* o1, float* o2, float* o3) {
    const int kp = blockIdx.x, head = blockIdx.y, c = threadIdx.x;
    float* out = head == 0 ? o0 : (head == 1 ? o1 : (head == 2 ? o2 : o3));
    const int src = sq16_l1_src(mode, kp);
    out[kp * kFcH + c] = src >= 0 ? net.w1raw[head][src * kFcH + c] : 0.0f;  // w1raw: rows in the ORIGINAL input order (w[3 head] may be permuted)
}

// The observation of a state, twice: the 48 row-words (memory | example_input | input_after | example_output | output_after,
// subleq.py:679-707) as 5-bit codes for the novelty hash, and the observation bits themselves (bit 5 w + j = bit j of row-word w's
// code, subleq.py:88-98) in layer 1's packed input order (sq16_l1_src) as FP8 bytes in the operand order of the MFMA lanes (QpMisc::obytes).
// Lane w < 48 computes the code of row-word w; a packed input is fetched from the lane that holds its word (ds_bpermute) and stored as its byte.
// The 16-byte words of a row of QpMisc::obytes are stored at position (word ^ swz(row)): the rows are 256 bytes = 64 banks apart, so the forward's
// reads of one word from all 16 rows would hit the same banks 16 times; swz = bits 8..11 of the row's LDS address (one more per row).
__device__ __forceinline__ int qp_ob_swz(const uint8_t* row) { return static_cast<int>((reinterpret_cast<uintptr_t>(row) >> 8) & 15u); }
__device__ __forceinline__ void qp_emit_obs(uint8_t* words, uint8_t* ob, unsigned long long mem, unsigned long long exin, unsigned long long inaf,
                                            unsigned long long exout, unsigned long long outaf, int lane, const int l1_mode) {
    uint32_t code = 0u;
    if (lane < 16) {
        code = static_cast<uint32_t>(mem >> (4 * lane)) & 15u;
    } else if (lane < 48) {
        const int row = (lane - 16) >> 3, sh = 8 * (lane & 7);
        const unsigned long long src = row == 0 ? exin : (row == 1 ? inaf : (row == 2 ? exout : outaf));
        code = qp_code(static_cast<uint32_t>(src >> sh) & 0xFFu);
    }
    if (lane < 48) words[lane] = static_cast<uint8_t>(code);
    const int swz = qp_ob_swz(ob);
    const int np = l1_mode == 0 ? 4 : 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p >= np) break;
        const int kp = 64 * p + lane;  // packed input: byte 64 (kp & 3) + (kp >> 2) of the row
        int w = 0, j = 0;
        bool valid = true;
        if (p == 0) {
            w = lane >> 2; j = lane & 3;
        } else {
            const int q = kp - 64;
            const int wq = (q * 52429) >> 18;  // q / 5 for q < 2^14
            j = q - 5 * wq;
            if (l1_mode == 0) { w = 16 + wq; valid = q < 160; }
            else { w = 24 + wq + (wq >= 8 ? 8 : 0); valid = q < 80; }
        }
        const uint32_t c = static_cast<uint32_t>(__shfl(static_cast<int>(code), valid ? w : 0));
        const int idx = 64 * (kp & 3) + (kp >> 2);
        ob[(((idx >> 4) ^ swz) << 4) | (idx & 15)] = (valid && ((c >> j) & 1u) != 0) ? static_cast<uint8_t>(0x38) : static_cast<uint8_t>(0);
    }
}

// One half-round of the forward's weight stream (search_persist.hpp ps_round, in halves): 8 k-steps of four MFMAs each on the prefetched weights
// Bq[8 PH .. 8 PH + 7], every k-step followed by the refill of its registers with the k-step the same registers serve next -- the next half-round
// of the same phase PH, 16 k-steps ahead in the stream (`rs`, `sbase`).  The stream W1(h) | W2(h) | W1(h + 1) ... is 7 (or 5) + 8 half-rounds per
// head, an odd number, so the phases alternate from head to head (static: the head loop is unrolled) and after the third head one phase is
// skipped (its registers were refilled half a round early; nothing is issued for it), which makes every forward start in phase 0.
// ABITS: layer 1, the A operands of the 8 k-steps are the 8 FP8 bytes of one uint2 of QpMisc::obytes, one v_cvt_pk_f32_fp8 converts two of
// them (every VALU instruction between the MFMAs of this loop costs ~18 clocks: with a v_bfe + v_cvt pair per k-step on the observation's
// bits a layer-1 round of 16 k-steps took 2 650 clocks against 2 176 for a layer-2 round, whose operand is a ds_read); otherwise layer 2, A
// operands from the H1 tile kPsARing k-steps ahead.  CHAIN 1 / 2: the half-round also carries dependent MFMAs 0..7 / 8..15 of the previous
// head's layer-3 quarter chain (A from the H2 columns this wave wrote, B = `w3` in registers).
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ floatx2 qp_abytes2(const uint2& w, const int i /* even k-step of the half-round */) {
    const uint32_t d = (i >> 2) == 0 ? w.x : w.y;
    return (i & 2) ? __builtin_amdgcn_cvt_pk_f32_fp8(static_cast<int>(d), true) : __builtin_amdgcn_cvt_pk_f32_fp8(static_cast<int>(d), false);
}
constexpr int kQpAAhead = 2;  // layer 1: a pair of A operands is converted this many k-steps before the MFMAs that read them (ring av[])
constexpr int kQpHalf = 8;    // k-steps of a half-round
#ifndef EAZ_QP_SB_MASK
#define EAZ_QP_SB_MASK 0
#endif
template <int CHAIN, bool ABITS, int PH>
__device__ __forceinline__ void qp_half(floatx4 (&acc)[4], floatx4 (&Bq)[kPsDepth], float (&av)[kPsARing], const float* apn, const uint2 abits,
                                        const uint2 abits_next, __amdgpu_buffer_rsrc_t rs, int voff, int sbase, floatx4& a3, const float* hpA,
                                        const float (&w3)[16]) {
    static_assert(kPsDepth == 2 * kQpHalf && kPsARing == kQpHalf && kQpAAhead < kPsARing && kChRing <= kQpHalf, "two half-rounds fill the weight ring; the operand ring is one half-round");
    constexpr int C0 = CHAIN == 2 ? 8 : 0;
    float cha[kChRing];
    if constexpr (CHAIN != 0) {
#pragma unroll
        for (int qq = 0; qq < kChRing; ++qq) cha[qq] = hpA[68 * (C0 + qq)];
    }
#pragma unroll
    for (int i = 0; i < kQpHalf; ++i) {
        // layer 1: av[] holds the operands of k-steps i .. i + kQpAAhead - 1 on entry (of this half-round; the last k-steps fill in the next one's)
        const float aop = av[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, Bq[kQpHalf * PH + i][c], acc[c], 0, 0, 0);
        if constexpr (CHAIN != 0) {
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(cha[i % kChRing], w3[C0 + i], a3, 0, 0, 0);
            if (i + kChRing < kQpHalf) cha[i % kChRing] = hpA[68 * (C0 + i + kChRing)];
        }
        Bq[kQpHalf * PH + i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + i * 4096, 0));
        if constexpr (ABITS) {
            if ((i & 1) == 0) {
                const floatx2 f = i + kQpAAhead < kQpHalf ? qp_abytes2(abits, i + kQpAAhead) : qp_abytes2(abits_next, i + kQpAAhead - kQpHalf);
                av[(i + kQpAAhead) % kPsARing] = f[0];
                av[(i + kQpAAhead + 1) % kPsARing] = f[1];
            }
        } else {
            av[i] = apn[i * 68];
        }
        __builtin_amdgcn_sched_barrier(EAZ_QP_SB_MASK);  // keep every refill right behind the k-step that freed its registers
    }
}
// the same with the phase as an argument (a constant once the head loop is unrolled)
template <int CHAIN, bool ABITS>
__device__ __forceinline__ void qp_half_ph(const int ph, floatx4 (&acc)[4], floatx4 (&Bq)[kPsDepth], float (&av)[kPsARing], const float* apn, const uint2 abits,
                                           const uint2 abits_next, __amdgpu_buffer_rsrc_t rs, int voff, int sbase, floatx4& a3, const float* hpA,
                                           const float (&w3)[16]) {
    if (ph == 0) qp_half<CHAIN, ABITS, 0>(acc, Bq, av, apn, abits, abits_next, rs, voff, sbase, a3, hpA, w3);
    else qp_half<CHAIN, ABITS, 1>(acc, Bq, av, apn, abits, abits_next, rs, voff, sbase, a3, hpA, w3);
}

// pgx.Env.step (A.1) around subleq.py:648-677 for ONE environment on one wavefront, compact state `src` (LDS) -> `dst` (LDS), everything
// wavefront-uniform.  mode 0: plain step (search), 1: auto_reset(env.step, env.init) (selfplay.py:26-75), as subleq16_wave_body.
// blob: also the 336-byte state of include/eaz.h (real steps); words: the observation's row-words.  Returns the reward / terminated flag.
__device__ __forceinline__ void qp_env_step(const EnvDev& e, const uint32_t* src, uint32_t* dst, const uint32_t* root_ex, uint8_t* words,
                                            uint8_t* xb, const int action, const int mode, const Key envkey, const int b, const int lane, char* blob0,
                                            char* blob1, float& reward_out, int& term_out, int16_t* tree_task, const int l1_mode,
                                            uint8_t* l1var, unsigned long long* dstamp = nullptr) {
    EAZ_QESTAMP(dstamp, 8);
    uint32_t s[QS_WORDS];
#pragma unroll
    for (int i = 0; i < QS_WORDS; ++i) s[i] = __builtin_amdgcn_readfirstlane(src[i]);
    unsigned long long mem = qp_u64(s[QS_MEM], s[QS_MEM + 1]);
    unsigned long long inaf = qp_u64(s[QS_INAF], s[QS_INAF + 1]), outaf = qp_u64(s[QS_OUTAF], s[QS_OUTAF + 1]);
    const uint32_t hdr = s[QS_HDR];
    const int h_step = hdr & 0xFF, s_term = (hdr >> 8) & 1, s_trunc = (hdr >> 9) & 1, s_solved = (hdr >> 10) & 1;
    int exroot = (hdr >> 11) & 1;
    // ---- decide what this environment does (pgx wrapper A.1, auto_reset selfplay.py:66-71, _step subleq.py:671-677)
    const bool done = (s_term | s_trunc) != 0;
    const bool do_init = mode == 1 && done;
    bool finish = false, run = false;
    int step_count, task = __builtin_amdgcn_readfirstlane(static_cast<int>(*tree_task));
    if (do_init) {
        const Key ki = fold(fold(envkey, static_cast<uint32_t>(b)), 1u);
        const int idx = e.n_tasks > 1 ? static_cast<int>(rbits(ki, 0) % static_cast<uint32_t>(e.n_tasks)) : 0;  // subleq.py:624
        task = __builtin_amdgcn_readfirstlane(e.tasks[idx]);
        if (lane == 0) *tree_task = static_cast<int16_t>(task);
        step_count = 0;
        run = true;
        mem = 0ull;
    } else if (done) {
        step_count = h_step;  // same state, zero rewards
    } else {
        step_count = h_step + 1;
        if (step_count >= 16 - 3 || s_solved) finish = true;  // subleq.py:671-673
        else run = true;
    }
    const int branch = task - 1 < 0 ? 0 : (task - 1 > 5 ? 5 : task - 1);  // get_test_cases: lax.switch clamps the branch index
    // the packed test vectors of case 0 (99.9 % of the steps run nothing else; the other two are fetched when case 0 has passed: twelve scalar
    // registers held across the interpreter were part of what the compiler spilled into VGPR lanes around it)
    const unsigned long long t_in = kSq16Cases.v[branch][0][0], t_out = kSq16Cases.v[branch][0][1];
    unsigned long long exin = exroot ? qp_u64(root_ex[0], root_ex[1]) : t_in, exout = exroot ? qp_u64(root_ex[2], root_ex[3]) : t_out;
    bool all_correct = false;
    int bytes_used = 0;
    int cost = 0;  // interpreted cycles of this step (all cases)
    int hint = 0;
    uint32_t touched = 0xFFFFu;
#ifdef EAZ_FC_STAMPS
    int dg_it0 = 0, dg_cases = 0;
#endif
    EAZ_QESTAMP(dstamp, 9);
    if (run) {
        const int idx = step_count - 1;  // the action is word step_count - 1 (subleq.py:654)
        if (!do_init && idx >= 0 && idx < 16) mem = (mem & ~(15ull << (4 * idx))) | (static_cast<unsigned long long>(action & 15) << (4 * idx));
        EAZ_QFSTAMP(dstamp, 0);
        // The runs whose results the parent state carries never fetched, read or wrote word idx: on the child's program they take the same
        // cycles to the same results whatever the word holds -- input_after / output_after stay, and the runs failed (a solved parent finishes,
        // :671-673, it does not run), so solved = 0 and the reward is 0.  Exact, and two thirds of the leaf steps of a search.
        const bool same_runs = !do_init && idx >= 0 && idx < 16 && ((qp_hdr_touched(hdr) >> idx) & 1u) == 0u;
        if (same_runs) {
            all_correct = false;
            hint = qp_hdr_hint(hdr);
            touched = qp_hdr_touched(hdr);
        } else {
            all_correct = true;
            unsigned tmask = 0u;
            asm volatile("" : "+v"(tmask));  // lives in a VGPR: subleq16_wave_run accumulates it on the vector port
#pragma unroll 1
            for (int c = 0; c < 3 && all_correct; ++c) {  // subleq.py:511 vmap; `solved` is the AND (:525), only case 0 feeds the state
                const unsigned long long sin = c == 0 ? t_in : kSq16Cases.v[branch][c][0], stout = c == 0 ? t_out : kSq16Cases.v[branch][c][1];
                unsigned long long s_after, s_out;
                int bytes_c;
                int it_c = 0;
#ifdef EAZ_FC_STAMPS
                if (c == 0) { asm volatile("" ::"s"(sin), "s"(stout)); EAZ_QFSTAMP(dstamp, 1); }
#endif
                const bool correct = subleq16_wave_run(mem, sin, stout, s_after, s_out, bytes_c, &it_c, &tmask);
#ifdef EAZ_FC_STAMPS
                if (c == 0) { asm volatile("" ::"s"(s_after), "s"(s_out), "s"(bytes_c)); EAZ_QFSTAMP(dstamp, 2); }
                if (c == 0) dg_it0 = it_c;
                dg_cases = c + 1;
#endif
                cost += it_c;
                bytes_used = max(bytes_used, bytes_c);  // :530
                if (c == 0) { inaf = s_after; outaf = s_out; }
                all_correct = correct;
            }
            hint = min(cost >> 2, 15);
            touched = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(tmask))) & 0xFFFFu;
        }
#ifdef EAZ_FC_STAMPS
        if (dstamp != nullptr && lane == 0) dstamp[15] = static_cast<unsigned long long>(cost) + 1ull;  // interpreted cycles of the stamped step (+ 1: "written")
        if (blockIdx.x < 64 && lane == 0 && mode == 0) {
            atomicAdd(&g_qp_ehist[dg_cases], 1ull);  // [0]: steps that needed no run
            atomicAdd(&g_qp_ehist[4 + min(cost >> 2, 31)], 1ull);
            atomicAdd(&g_qp_ehist[36 + min(dg_it0 >> 2, 31)], 1ull);
            if (dg_cases > 1) atomicAdd(&g_qp_ehist[68 + min(cost >> 2, 31)], 1ull);
            atomicAdd(&g_qp_ehist[100], static_cast<unsigned long long>(cost));
        }
#endif
        exin = t_in; exout = t_out; exroot = 0;
    }
    EAZ_QESTAMP(dstamp, 10);
    float reward;
    int terminated, solved;
    if (run) {
        solved = all_correct ? 1 : 0;                           // subleq.py:525
        reward = solved ? subleq_reward(e, bytes_used) : 0.0f;  // :535-537 / :540-542
        terminated = do_init ? 0 : s_term;                      // _step never sets terminated on this branch (:661-669)
    } else {
        solved = s_solved;
        reward = 0.0f;                                          // terminal no-op (A.1) or subleq.py:673
        terminated = finish ? 1 : s_term;
    }
    const int truncated = do_init ? 0 : s_trunc;
    if (lane == 0) {
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        d4[0] = make_uint4(static_cast<uint32_t>(mem), static_cast<uint32_t>(mem >> 32), static_cast<uint32_t>(inaf), static_cast<uint32_t>(inaf >> 32));
        d4[1] = make_uint4(static_cast<uint32_t>(outaf), static_cast<uint32_t>(outaf >> 32), qp_hdr(step_count, terminated, truncated, solved, exroot, hint, touched),
                           __builtin_bit_cast(uint32_t, reward));
    }
    EAZ_QESTAMP(dstamp, 11);
    if (lane == 0) *l1var = static_cast<uint8_t>(exroot);
    qp_emit_obs(words, xb, mem, exin, inaf, exout, outaf, lane, l1_mode);
    EAZ_QESTAMP(dstamp, 12);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        char* blob = k == 0 ? blob0 : blob1;
        if (!blob) continue;
        // the 336-byte blob of include/eaz.h: header, four rows of eight uint16, 256 memory bytes (zero beyond the 16 words)
        if (lane == 0)
            *reinterpret_cast<uint4*>(blob) = make_uint4(static_cast<uint32_t>(step_count), static_cast<uint32_t>(task), __builtin_bit_cast(uint32_t, reward),
                                                         static_cast<uint32_t>(terminated) | (static_cast<uint32_t>(truncated) << 8) | (static_cast<uint32_t>(solved) << 16));
        if (lane < 32) {
            const int row = lane >> 3, sh = 8 * (lane & 7);
            const unsigned long long srcv = row == 0 ? exin : (row == 1 ? inaf : (row == 2 ? exout : outaf));
            reinterpret_cast<uint16_t*>(blob + 16)[lane] = static_cast<uint16_t>((srcv >> sh) & 0xFFull);
        }
        if (lane < 16) {
            uint4 c = make_uint4(0u, 0u, 0u, 0u);
            if (lane == 0) {
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t n4 = static_cast<uint32_t>(mem >> (16 * i)) & 0xFFFFu;
                    w[i] = (n4 & 0xFu) | ((n4 & 0xF0u) << 4) | ((n4 & 0xF00u) << 8) | ((n4 & 0xF000u) << 12);
                }
                c = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4*>(blob + 80 + 16 * lane) = c;
        }
    }
    reward_out = reward;
    term_out = terminated;
    EAZ_QESTAMP(dstamp, 13);
}

// novelty of the 16 observations whose row-words are in M.words (XXHash variant, hashes.py:162-229: lane = (environment, stripe); the IO
// rows or, subleq_hash_only_io = False, the whole observation: sq16_words_hash_index), one wavefront
__device__ __forceinline__ void qp_novelty(const NetDev& net, QpMisc& M, const int lane) {
    const int ev = lane >> 2, i = lane & 3;
    const uint32_t index = sq16_words_hash_index(&M.words[ev][0], i, net.hash_off == 0, net.hash_bits);
    if (i == 0) {
        const bool seen = net.bitset ? ((net.bitset[index >> 3] >> (index & 7u)) & 1u) != 0 : false;  // hashes.py:29-38
        M.o_nov[ev] = (seen ? 0.0f : 1.0f) * net.nov_scale;                                            // fully_connected.py:90
    }
}

// Layer-1 mode 1 (the network's input order puts example_input / example_output first, sq16_l1_src): those 80 inputs are constants of a
// tree between two resets, so the chain over them -- acc = +0; acc = fmaf(x[k], W1[k][c], acc) for k = 80..119, 160..199, the first 80 steps
// of the oracle's chain under that order -- is computed once and every forward starts from it.  A zero input leaves the accumulator as it
// is (exact), a one adds its weight row: fmaf(1, w, acc).  The vectors of the six task branches (get_test_cases clamps, subleq.py:459-501)
// are shared by all trees: kQpL1Tab floats at the start of `l1init`, filled by sq16_l1_table_kernel; a tree's root blob may carry other rows
// (a hand-made state), whose vector the prologue writes behind the table.
constexpr int kQpL1Tab = 6 * 3 * kFcH;
__device__ __forceinline__ float qp_l1_chain(const float* W /* column of the head's W1, original input order */, const unsigned long long exin,
                                             const unsigned long long exout) {
    float acc = 0.0f;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const unsigned long long row = half == 0 ? exin : exout;
        const int k0 = half == 0 ? 80 : 160;  // observation inputs of example_input / example_output (subleq.py:697-705)
#pragma unroll 1
        for (int w = 0; w < 8; ++w) {
            const uint32_t code = qp_code(static_cast<uint32_t>(row >> (8 * w)) & 0xFFu);
#pragma unroll
            for (int j = 0; j < 5; ++j)
                if ((code >> j) & 1u) acc = __builtin_fmaf(1.0f, W[(k0 + 5 * w + j) * kFcH], acc);
        }
    }
    return acc;
}
// grid = (6 task branches, 3 heads), 256 threads = hidden columns
__global__ void __launch_bounds__(256) sq16_l1_table_kernel(NetDev net, HeadList heads, float* l1init) {
    const int branch = blockIdx.x, head = blockIdx.y, col = threadIdx.x;
    l1init[(branch * 3 + head) * kFcH + col] = qp_l1_chain(net.w1raw[heads.id[head]] + col, kSq16Cases.v[branch][0][0], kSq16Cases.v[branch][0][1]);
}
// the root blobs' vectors of the workgroup's 16 trees: thread -> (tree, head, column)
__device__ __forceinline__ void qp_l1_start_roots(const NetDev& net, const HeadList& heads, float* l1root_wg /* [16][3][256] */, const QpMisc& M, const int tid) {
    for (int it = tid; it < kQpTrees * 3 * kFcH; it += kQpThreads) {
        const int col = it & (kFcH - 1), head = (it >> 8) % 3, ti = it / (3 * kFcH);
        l1root_wg[it] = qp_l1_chain(net.w1raw[heads.id[head]] + col, qp_u64(M.root_ex[ti][0], M.root_ex[ti][1]), qp_u64(M.root_ex[ti][2], M.root_ex[ti][3]));
    }
}

// The compact state of this kernel (QS_*) holds less than the 336-byte blob of include/eaz.h: program words as 4 bits (and none beyond the
// sixteenth), IO values as bytes with 16 as "beyond the valid ones", the step count in 8 bits, the task in 16.  Every state the environment
// itself produces fits (subleq.py:88-98,648-677); a caller's hand-made blob may not, and would be truncated silently.  The host runs this
// check on the blobs a call starts from and sends a batch with any misfit to the per-simulation kernels, which keep whole blobs.
__global__ void __launch_bounds__(256) sq16_roots_fit_kernel(const char* __restrict__ states, long stride, int B, int* misfit, int epoch) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const uint4* p = reinterpret_cast<const uint4*>(states + static_cast<long>(b) * stride);
    const uint4 hv = p[0];
    const int step = static_cast<int>(hv.x), task = static_cast<int>(hv.y);
    bool ok = step >= 0 && step <= 255 && task >= -32768 && task <= 32767;
#pragma unroll
    for (int i = 1; i <= 4; ++i) {  // four rows of eight uint16: every value <= 16
        const uint4 r = p[i];
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) ok = ok && (w[k] & 0xFFFFu) <= 16u && (w[k] >> 16) <= 16u;
    }
    const uint4 m0 = p[5];  // the sixteen program words: each < 16
    ok = ok && ((m0.x | m0.y | m0.z | m0.w) & 0xF0F0F0F0u) == 0u;
    uint32_t tail = 0u;
    for (int i = 6; i < 21; ++i) { const uint4 m = p[i]; tail |= m.x | m.y | m.z | m.w; }  // bytes 16..255 of the memory: zero
    ok = ok && tail == 0u;
    if (!ok) *misfit = epoch;  // `misfit` is host memory mapped into the device's address space; the host compares it with this call's epoch
}

// evaluate (PersistSelfplay::on == 2), after the launch: the reference's while_loop (evaluate.py:13-57) steps EVERY environment until the last
// one has terminated; a workgroup stops when its own 16 have.  For an environment that is already terminal a further env.step returns the
// same state with zero rewards (pgx wrapper A.1), so the only trace of the iterations a workgroup did not run is the blob's reward field
// (every state blob starts with {int32 step_count; int32 aux; float reward0; ...}, envs.hpp StateHeader).
__global__ void __launch_bounds__(256) eval_finalize_kernel(char* states, long stride, int B, int trees_per_wg, const int* wg_iters, const int* iters) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    if (wg_iters[b / trees_per_wg] < *iters) reinterpret_cast<StateHeader*>(states + static_cast<long>(b) * stride)->reward0 = 0.0f;
}

// The kernel's arguments live in the kernarg segment and every phase reads them through a pointer the compiler cannot see through (an
// empty asm redefines it per phase): a by-value struct has all its loads hoisted to the kernel entry as loop invariants, and the SGPRs
// the phases need between them then spill through VGPR lanes (first build: 193 SGPR spills).  Re-issued scalar loads hit the scalar
// cache.  Sq16SelfplayArgs is the kernel's ONLY parameter, so it sits at offset 0 of the kernarg segment (tests/test_isa_guard.py).
typedef const Sq16SelfplayArgs* QpArgsPtr;
__device__ __forceinline__ QpArgsPtr qp_args() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const Sq16SelfplayArgs*)p;  // constant -> generic; the loads are inferred back to scalar loads after inlining
}
#define EAZ_QP_ARGS                          \
    const Sq16SelfplayArgs& a = *qp_args();  \
    const TreeDev& t = a.t

// EVAL: the evaluate mode (PersistSelfplay::on == 2) is its own instantiation, so that self-play and search pay nothing for it (as a run-time
// flag it cost the C3 launch 1.7 %)
template <bool EVAL>
__global__ void __launch_bounds__(kQpThreads) subleq16_selfplay_kernel(Sq16SelfplayArgs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    int N, S, B, n_steps, max_depth;
    bool search_only;
    constexpr bool evaluate = EVAL;   // src/evaluate.py:13-57 as a device-side loop (PersistSelfplay::on == 2)
    float* logits_base;
    {
        EAZ_QP_ARGS;
        search_only = a.sp.on == 0;
        N = t.N; S = t.S; B = t.B; n_steps = search_only ? 1 : a.sp.T; max_depth = t.max_depth;
        logits_base = a.logits;
    }
    const QpLds L = qp_lds(N, S);
    float* H1 = reinterpret_cast<float*>(lds + L.off_h1);
    float* H2 = reinterpret_cast<float*>(lds + L.off_h2);
    float* part = reinterpret_cast<float*>(lds + L.off_part);
    uint32_t* trees = reinterpret_cast<uint32_t*>(lds + L.off_tree);
    uint32_t* states = reinterpret_cast<uint32_t*>(lds + L.off_state);
    uint16_t* path = reinterpret_cast<uint16_t*>(lds + L.off_path);  // [16][S]: node | action << 8
    QpMisc& M = *reinterpret_cast<QpMisc*>(lds + L.off_misc);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int b0 = blockIdx.x * kQpTrees;
    // Every phase derives its lane roles from a thread id the compiler cannot trace to threadIdx (an empty asm): otherwise the ~40 registers
    // of lane-derived pointers / indices of ALL phases are computed once at the kernel entry and held (or spilled) across the whole loop.
    auto phase_tid = [&]() { int t_ = tid; asm volatile("" : "+v"(t_)); return t_; };
    // tree lanes: 32 per tree (a tree never leaves its wavefront); re-score: slot = node of the path, q = quarter of the actions.  Trees
    // beyond the batch mirror the last environment; they never store outside LDS / their own scratch rows.
#define EAZ_QP_TREE_LANES                                                                                            \
    const int pt_ = phase_tid();                                                                                     \
    const int ti = pt_ >> 5, tl = pt_ & 31, slot = tl >> 2, q = tl & 3;                                              \
    uint32_t* T = trees + ti * L.tstride;                                                                            \
    uint32_t* ST = states + ti * N * QS_WORDS;                                                                       \
    uint16_t* P = path + ti * S;                                                                                     \
    const bool tree_live = b0 + ti < B;                                                                              \
    const int gb = tree_live ? b0 + ti : B - 1;                                                                      \
    float* LG = logits_base + static_cast<long>(b0 + ti) * N * (2 * kQpA);                                           \
    const Group<4, 4> g = make_group<4, 4>(gb, q, kQpA);                                                             \
    const float beta = M.beta[ti];                                                                                   \
    const bool need_var = beta != 0.0f;                                                                              \
    (void)ST; (void)P; (void)LG; (void)slot; (void)need_var; (void)tree_live
    // forward lanes: lane (r, kq) of wave w owns hidden columns 64 w + 4 r .. + 3
#define EAZ_QP_FWD_LANES                                                           \
    const int pf_ = phase_tid() & 63;                                              \
    const int fr = pf_ & 15, fkq = pf_ >> 4, fw = wave & 3;                        \
    const int fn0 = fw * 64, fcol = fn0 + 4 * fr, fvoff = (fkq * kFcH + fcol) * 4; \
    (void)fcol

    // ---------------------------------------------------------------- prologue ----------------------------------------------------
    floatx4 Bq[kPsDepth];  // the weight stream, 16 k-steps ahead
    {
        EAZ_QP_ARGS;
        EAZ_QP_FWD_LANES;
        for (int i = tid; i < 3 * 2 * kFcH; i += kQpThreads) {
            const int h = i / (2 * kFcH), l = (i / kFcH) & 1, c = i & (kFcH - 1);
            M.b12[h][l][c] = a.net.b[a.heads.id[h] * 3 + l][c];
        }
        if (tid < 18) M.b3[tid] = tid < 2 ? a.net.b[a.heads.id[tid] * 3 + 2][0] : a.net.b[a.heads.id[2] * 3 + 2][tid - 2];
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.net.w1q[a.heads.id[0]]), 0, a.net.l1q_halves * kQpHalf * 4096, 0x00020000);
        if (wave < 4) {
#pragma unroll
            for (int i = 0; i < kPsDepth; ++i) Bq[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs0, fvoff, i * 4096, 0));
        }
    }
    for (int k = 0; k < 2; ++k) {
        EAZ_QP_ARGS;
        // the current states of two of the workgroup's environments: 336-byte blob -> compact root state + the observation's row-words
        const int j = 2 * wave + k;
        const int bj = b0 + j < B ? b0 + j : B - 1;
        // self-play: the caller's state batch; search only: the root embedding tree_root_kernel wrote
        const char* sp = search_only ? t.embeddings + static_cast<long>(bj) * N * sizeof(eaz_subleq_state)
                                     : a.sp.states + static_cast<long>(bj) * sizeof(eaz_subleq_state);
        const uint4 hv = *reinterpret_cast<const uint4*>(sp);
        const uint4 r0 = *reinterpret_cast<const uint4*>(sp + 16), r1 = *reinterpret_cast<const uint4*>(sp + 32);
        const uint4 r2 = *reinterpret_cast<const uint4*>(sp + 48), r3 = *reinterpret_cast<const uint4*>(sp + 64);
        const uint4 mc = *reinterpret_cast<const uint4*>(sp + 80);
        // rows are eight uint16 = two uint4 halves: ex_in = (r0), ex_in_after = (r1), ex_out = (r2), ex_out_after = (r3) -- 16 bytes each
        auto row = [](const uint4& r) { return qp_row_bytes(r); };
        const unsigned long long mem = qp_nibbles(mc), exin = row(r0), inaf = row(r1), exout = row(r2), outaf = row(r3);
        if (lane == 0) {
            uint32_t* rs = M.root_state[j];
            rs[QS_MEM] = static_cast<uint32_t>(mem); rs[QS_MEM + 1] = static_cast<uint32_t>(mem >> 32);
            rs[QS_INAF] = static_cast<uint32_t>(inaf); rs[QS_INAF + 1] = static_cast<uint32_t>(inaf >> 32);
            rs[QS_OUTAF] = static_cast<uint32_t>(outaf); rs[QS_OUTAF + 1] = static_cast<uint32_t>(outaf >> 32);
            rs[QS_HDR] = qp_hdr(static_cast<int>(hv.x), hv.w & 0xFF, (hv.w >> 8) & 0xFF, (hv.w >> 16) & 0xFF, 1);  // touched: all ones (unknown)
            M.task[j] = static_cast<int16_t>(static_cast<int>(hv.y));
            rs[QS_REWARD] = hv.z;
            M.root_ex[j][0] = static_cast<uint32_t>(exin); M.root_ex[j][1] = static_cast<uint32_t>(exin >> 32);
            M.root_ex[j][2] = static_cast<uint32_t>(exout); M.root_ex[j][3] = static_cast<uint32_t>(exout >> 32);
            M.l1var[j] = 1;  // the root's example rows are the blob's
            M.sum_reward[j] = 0.0f;
            M.real_term[j] = static_cast<uint8_t>(hv.w & 0xFF);
        }
        qp_emit_obs(M.words[j], M.obytes[j], mem, exin, inaf, exout, outaf, lane, a.net.l1q_mode);
    }
    {
        // per-tree scalars of the whole launch; search only: the root the caller prepared (eaz_search_begin -> tree_root_kernel: masked prior
        // logits, value, variance, beta, scaled Gumbel noise, invalid actions) into an empty tree (mctx instantiate_tree_from_root, A.3)
        EAZ_QP_ARGS;
        const int pt_ = phase_tid();
        const int ti = pt_ >> 5, tl = pt_ & 31;
        const int gb = b0 + ti < B ? b0 + ti : B - 1;
        uint32_t* T = trees + ti * L.tstride;
        if (!search_only) {
            if (tl == 0) {
                M.beta[ti] = evaluate ? a.sp.sbeta * 1.0f  // evaluate.py:34 beta * ones
                                      : a.sp.sbeta * (B > 1 ? static_cast<float>(gb) / static_cast<float>(B - 1) : 0.0f);  // selfplay.py:105
                M.nc[ti] = kQpA < t.max_considered ? kQpA : t.max_considered;  // every action is legal (subleq.py:626)
            }
            if (tl < 16) M.invalid[ti][tl] = 0;
        } else {
            for (int n = tl; n < N; n += 32) {
                uint32_t* R = T + n * QN_WORDS;
#pragma unroll
                for (int w = 0; w < QN_WORDS; ++w) R[w] = w >= QN_SEL ? 0xFFFFFFFFu : 0u;
            }
            ps_wave_sync();
            const long nb = static_cast<long>(gb) * N;
            float lgv = -kInf;
            int inv = 1;
            if (tl < 16) {
                lgv = t.children_prior_logits[nb * kQpA + tl];
                inv = t.root_invalid[static_cast<long>(gb) * kQpA + tl] != 0;
                M.gumbel[ti][tl] = t.root_gumbel[static_cast<long>(gb) * kQpA + tl];
                M.invalid[ti][tl] = static_cast<uint8_t>(inv);
            }
            const int nvalid = gsumi<16>(tl < 16 ? 1 - inv : 0);
            const float pm = gmax<16>(lgv);
            const float pe = tl < 16 ? det_exp(lgv - pm) : 0.0f;
            const float ps = gsum<16>(pe);
            if (tl < 16) { M.root_logits[ti][tl] = lgv; M.root_probs[ti][tl] = pe / ps; }
            if (tl == 0) {
                M.beta[ti] = t.beta[gb];
                M.nc[ti] = nvalid < t.max_considered ? nvalid : t.max_considered;
                T[QN_RAW_VALUE] = f2u(t.raw_values[nb]); T[QN_NODE_VALUE] = f2u(t.node_values[nb]);
                T[QN_RAW_VAR] = f2u(t.raw_var[nb]); T[QN_NODE_VAR] = f2u(t.node_var[nb]);
                T[QN_META] = static_cast<uint32_t>(t.node_visits[nb] & 0xFF) | (0xFFu << 8);
                M.D[ti] = 0; M.parent[ti] = 0; M.action[ti] = 0; M.dst[ti] = 0;
            }
        }
    }
    __syncthreads();
    {
        EAZ_QP_ARGS;
        if (a.net.l1q_mode == 1) {
            qp_l1_start_roots(a.net, a.heads, a.l1init + kQpL1Tab + static_cast<long>(b0) * (3 * kFcH), M, phase_tid());
            __syncthreads();  // global stores -> loads of other waves of the workgroup (one CU, one L1)
        }
    }
    if (search_only) {
        EAZ_QP_TREE_LANES;
        if (tl < QS_WORDS) ST[tl] = M.root_state[ti][tl];  // embedding of node 0
    }

    // output `slot` (0 value, 1 ube, 2.. policy logits) of row `row` from the four quarter sums: ((c0 + c1) + (c2 + c3)) + b (o_linear_parts)
    auto head_out = [&](int row, int slot) {
        const float* pp = part + row * kQpPartLd + slot;
        const float c0 = pp[0], c1 = pp[16 * kQpPartLd], c2 = pp[32 * kQpPartLd], c3 = pp[48 * kQpPartLd];
        return ((c0 + c1) + (c2 + c3)) + M.b3[slot];
    };
    const int n_iter = n_steps * (S + 1);
    // evaluate: cond_fn (evaluate.py:18-20) for this workgroup's environments -- every thread reads the same 16 flags
    auto all_terminated = [&]() {
        bool all = true;
        for (int j = 0; j < kQpTrees; ++j) all = all && (b0 + j >= B || M.real_term[j] != 0);
        return all;
    };
    int steps_done = n_steps;  // real steps this workgroup ran (evaluate: its trip count)
    if (evaluate && (all_terminated() || n_steps <= 0)) steps_done = 0;
    floatx4 pre_lg = floatx4{0.0f, 0.0f, 0.0f, 0.0f}, pre_pr = floatx4{0.0f, 0.0f, 0.0f, 0.0f};  // prefetched logits / probabilities of path level `slot`
    int pre_node = -1;
#pragma unroll 1
    for (int it = search_only ? 1 : 0; it <= n_iter && steps_done > 0; ++it) {  // search only: no root forward, the caller evaluated the root
        const int step = it / (S + 1), s = it - step * (S + 1);  // s = 0: root of real step `step`; s >= 1: simulation s - 1
        int env_mode = 0;
        Key env_key{0u, 0u};
        if (s >= 2 || (s == 0 && it > 0)) {
            // ---- expand + backward of the simulation evaluated last (simulation s - 2; at s == 0: simulation S - 1 of the step before) ----
            EAZ_QP_ARGS;
            EAZ_QP_TREE_LANES;
            EAZ_QPSTAMP(0);
            const int D = M.D[ti], dst = M.dst[ti], parent = M.parent[ti], action = M.action[ti];
            {
                // epistemic_recurrent_fn (src/context.py:132-153) on the forward's outputs; mctx update_tree_node + topology (A.3)
                const float z = tl < 18 ? head_out(ti, tl < 16 ? 2 + tl : tl - 16) : 0.0f;
                // one det_tanh region for the two one-output heads (lanes 16 and 17): as two `if`s the wave ran the sequence twice
                const float th = (tl == 16 || tl == 17) ? det_tanh(z) : 0.0f;
                if (tl == 16) M.o_value[ti] = th;                                                 // fully_connected.py:55
                if (tl == 17) M.o_ube[ti] = (0.5f * (th + 1.0f)) * a.net.max_u;                   // :64, :93
                float lgv = tl < 16 ? z : -kInf;
                const float m = gmax<16>(lgv);                                                    // :135
                ps_wave_sync();
                lgv = lgv - m;                                                                     // :137 (every action legal, subleq.py:626)
                // jax.nn.softmax of the node's prior logits, once per node (qtransform_p): Group::softmax on 16 lanes x 1 action -- the butterfly
                // sum over masks 1, 2, 4, 8 is the pairwise tree of lsum<4> + masks 1, 2 of the 4 x 4 form, so the bits are the same
                const float pm = gmax<16>(lgv);
                const float pe = tl < 16 ? det_exp(lgv - pm) : 0.0f;
                const float ps = gsum<16>(pe);
                const float pp = pe / ps;
                if (tl < 16) {
                    M.new_logits[ti][tl] = lgv; M.new_probs[ti][tl] = pp;
                    LG[dst * (2 * kQpA) + tl] = lgv; LG[dst * (2 * kQpA) + kQpA + tl] = pp;
                }
                if (tl == 0) {
                    const bool term = M.o_term[ti] != 0;
                    const float value = term ? 0.0f : M.o_value[ti];                               // :140
                    const float var = term ? 0.0f : ube_combine(M.o_ube[ti], M.o_nov[ti], t.max_u);    // :141
                    uint32_t* Rn = T + dst * QN_WORDS;
                    Rn[QN_RAW_VALUE] = f2u(value); Rn[QN_NODE_VALUE] = f2u(value);
                    Rn[QN_RAW_VAR] = f2u(var); Rn[QN_NODE_VAR] = f2u(var);
                    Rn[QN_REWARD] = f2u(M.o_reward[ti]);                                           // :139
                    const uint32_t visits = (Rn[QN_META] & 0xFFu) + 1u;
                    Rn[QN_META] = visits | (static_cast<uint32_t>(parent) << 8) | (static_cast<uint32_t>(action) << 16) | (term ? 1u << 20 : 0u);  // :142-144
                    uint8_t* ch = reinterpret_cast<uint8_t*>(T + parent * QN_WORDS + QN_CHILD);
                    ch[action] = static_cast<uint8_t>(dst);
                }
            }
            ps_wave_sync();
            // ---- backward (A.6): lane = level of the recorded path ----
            const float disc_live = t.two_player ? (1.0f * t.discount) * -1.0f : 1.0f * t.discount;       // context.py:142-143
            const float leaf_v = u2f(T[dst * QN_WORDS + QN_RAW_VALUE]), leaf_u = u2f(T[dst * QN_WORDS + QN_RAW_VAR]);
            float Gin = leaf_v, Uin = leaf_u;
            if (tl < 16) {
                const int tj = tl;
                for (int c = (D - 1) >> 4; c >= 0; --c) {
                    const int d = c * 16 + tj;
                    const bool valid = d < D;
                    const int pn = P[valid ? d : 0] & 0xFF;
                    const int cn = (valid && d + 1 < D) ? (P[d + 1] & 0xFF) : dst;  // the child this level's edge leads to
                    uint32_t* R = T + pn * QN_WORDS;
                    const uint32_t* Cn = T + cn * QN_WORDS;
                    const float er = u2f(Cn[QN_REWARD]), ed = (Cn[QN_META] >> 20) & 1u ? 0.0f : disc_live;
                    const int cnt_i = static_cast<int>(R[QN_META] & 0xFFu);
                    const float nv = u2f(R[QN_NODE_VALUE]), nvar = u2f(R[QN_NODE_VAR]);
                    const int top = (D - c * 16 < 16 ? D - c * 16 : 16) - 1;  // deepest level of this chunk
                    // deepest level of the two trees of this wavefront: levels beyond it are skipped by a scalar branch (search_persist.hpp)
                    int wtop = __builtin_amdgcn_readlane(top, 0);
                    { const int t1 = __builtin_amdgcn_readlane(top, 32); wtop = wtop > t1 ? wtop : t1; }
                    *reinterpret_cast<float2*>(&M.edge[ti][tj][0]) = make_float2(er, ed);
                    ps_wave_sync();
                    float ers[16], eds[16];
#pragma unroll
                    for (int s4 = 0; s4 < 8; ++s4) {
                        const floatx4 v = *reinterpret_cast<const floatx4*>(&M.edge[ti][2 * s4][0]);  // wide broadcast reads
                        ers[2 * s4] = v[0]; eds[2 * s4] = v[1]; ers[2 * s4 + 1] = v[2]; eds[2 * s4 + 1] = v[3];
                    }
                    float G = Gin, U = Uin;
#pragma unroll
                    for (int lv = 15; lv >= 0; --lv) {
                        if (lv > wtop) continue;
                        const float rs = ers[lv], dd = eds[lv];
                        if (lv <= top && lv >= tj) {
                            G = rs + dd * G;
                            U = 0.0f + (dd * dd) * U;  // children_rewards_epistemic_variance == 0 (context.py:149)
                        }
                    }
                    const float cnt = static_cast<float>(cnt_i);
                    const float pv = (nv * cnt + G) / (cnt + 1.0f);
                    const float pvar = var_backup_update(t.var_backup, nvar, cnt, U);
                    if (valid) {
                        R[QN_NODE_VALUE] = f2u(pv);
                        R[QN_NODE_VAR] = f2u(pvar);
                        R[QN_META] = (R[QN_META] & ~0xFFu) | static_cast<uint32_t>(cnt_i + 1);
                    }
                    Gin = __shfl(G, 0, 16); Uin = __shfl(U, 0, 16);
                    ps_wave_sync();
                }
            }
            ps_wave_sync();
            EAZ_QPSTAMP(1);
        }
        if (search_only && it == n_iter) break;  // the last expand + backward is done: the tree goes back to the caller's arrays below
        if (s == 0 && it > 0) {
            // ---- the search of real step `step - 1` is complete: gumbel_muzero_policy tail (A.2, tree_finish_kernel), epistemic_summary
            // (A.7, tree_summary_kernel), action choice (selfplay.py:118-134), SelfplayOutput[t] except the state ----
            EAZ_QP_ARGS;
            EAZ_QP_TREE_LANES;
            const int pstep = step - 1;
            const Key kt = fold(a.sp.key, kTagStep + static_cast<uint32_t>(pstep));
            if (slot == 0) {
                NodeRows<4> rw;
                const uint32_t cw = T[QN_CHILD + q];
                float gum[4];
                uint8_t inv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (cw >> (8 * j)) & 0xFFu;
                    const bool has = c != 0xFF;
                    const uint32_t* Cn = T + (has ? c : 0) * QN_WORDS;
                    const uint32_t meta = Cn[QN_META];
                    inv[j] = M.invalid[ti][4 * q + j];
                    rw.logits[j] = M.root_logits[ti][4 * q + j];
                    rw.values[j] = has ? u2f(Cn[QN_NODE_VALUE]) : 0.0f;
                    rw.rewards[j] = has ? u2f(Cn[QN_REWARD]) : 0.0f;
                    rw.discounts[j] = has ? ((meta >> 20) & 1u ? 0.0f : (t.two_player ? (1.0f * t.discount) * -1.0f : 1.0f * t.discount)) : 0.0f;
                    rw.visits[j] = has ? static_cast<int>(meta & 0xFFu) : 0;
                    rw.index[j] = has ? c : -1;
                    rw.var[j] = has ? u2f(Cn[QN_NODE_VAR]) : 0.0f;
                    rw.rvar[j] = 0.0f;
                    gum[j] = M.gumbel[ti][4 * q + j];
                }
                float cq[4];
                {
                    const floatx4 pr4 = *reinterpret_cast<const floatx4*>(&M.root_probs[ti][4 * q]);
                    const float pr[4] = {pr4[0], pr4[1], pr4[2], pr4[3]};
                    qtransform_p<4, 4>(t, g, rw, pr, u2f(T[QN_RAW_VALUE]), u2f(T[QN_RAW_VAR]), beta, cq);
                }
                int lmax = 0, lsumN = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lmax = rw.visits[j] > lmax ? rw.visits[j] : lmax; lsumN += rw.visits[j]; }
                const int maxN = gmaxi<4>(lmax);
                const int totN = gsumi<4>(lsumN);
                int act = root_pick<4, 4>(t, g, rw, cq, static_cast<float>(maxN), gum, inv);
                float x[4], aw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = rw.logits[j] + cq[j];
                const float mx = g.vmax(x);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = inv[j] ? kF32Min : x[j] - mx;
                g.softmax(x, aw);
                const float ftot = static_cast<float>(totN);
                float qvar[4], vprob[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = rw.discounts[j];
                    qvar[j] = 0.0f + (d * d) * rw.var[j];  // children_rewards_epistemic_variance == 0
                    vprob[j] = static_cast<float>(rw.visits[j]) / (ftot > 1.0f ? ftot : 1.0f);
                }
                // action choice: categorical over visit PROBABILITIES / action weights used as logits (reproduced literally) = argmax(x + Gumbel)
                if (a.sp.flags & kSpSampleVisits) {
                    const Key kvis = fold(kt, kTagSampleVisits);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = vprob[j] + gumbel(kvis, static_cast<uint32_t>(gb * kQpA + 4 * q + j));
                    act = g.argmax(v);
                }
                if (a.sp.flags & kSpSampleImproved) {
                    const Key kimp = fold(kt, kTagSampleImproved);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = aw[j] + gumbel(kimp, static_cast<uint32_t>(gb * kQpA + 4 * q + j));
                    act = g.argmax(v);
                }
                const long o = static_cast<long>(pstep) * B + gb;
                if (tree_live && a.sp.out_qvar) *reinterpret_cast<floatx4*>(a.sp.out_qvar + o * kQpA + 4 * q) = floatx4{qvar[0], qvar[1], qvar[2], qvar[3]};
                if (q == 0) {
                    M.real_action[ti] = act;
                    if (tree_live) {
                        if (a.sp.out_value_pred) a.sp.out_value_pred[o] = M.root_v[ti];
                        if (a.sp.out_ube_pred) a.sp.out_ube_pred[o] = M.root_u[ti];
                        if (a.sp.out_root_value) a.sp.out_root_value[o] = u2f(T[QN_NODE_VALUE]);
                        if (a.sp.out_root_std) a.sp.out_root_std[o] = __builtin_sqrtf(u2f(T[QN_NODE_VAR]));
                        if (a.sp.out_action) a.sp.out_action[o] = act;
                    }
                }
            }
            env_mode = 1;
            env_key = fold(kt, kTagEnv);
        } else if (s >= 1) {
            // ---- re-score the nodes whose selection inputs changed: the path's levels 0 .. D-1 and the (re-)expanded node (s == 1: the root) ----
            EAZ_QP_ARGS;
            EAZ_QP_TREE_LANES;
            const int D = s >= 2 ? M.D[ti] : 0, dst = s >= 2 ? M.dst[ti] : 0;
            const int nl = D + 1;
            for (int l = slot; l < nl; l += 8) {
                const int node = l < D ? (P[l] & 0xFF) : dst;
                uint32_t* R = T + node * QN_WORDS;
                NodeRows<4> rw;
                const uint32_t cw = R[QN_CHILD + q];
                floatx4 lg4, pr4;
                if (node == 0) { lg4 = *reinterpret_cast<const floatx4*>(&M.root_logits[ti][4 * q]); pr4 = *reinterpret_cast<const floatx4*>(&M.root_probs[ti][4 * q]); }
                else if (l == D) { lg4 = *reinterpret_cast<const floatx4*>(&M.new_logits[ti][4 * q]); pr4 = *reinterpret_cast<const floatx4*>(&M.new_probs[ti][4 * q]); }
                else if (node == pre_node) { lg4 = pre_lg; pr4 = pre_pr; }  // requested when the select that recorded this path ended
                else {
                    lg4 = *reinterpret_cast<const floatx4*>(LG + node * (2 * kQpA) + 4 * q);
                    pr4 = *reinterpret_cast<const floatx4*>(LG + node * (2 * kQpA) + kQpA + 4 * q);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (cw >> (8 * j)) & 0xFFu;
                    const bool has = c != 0xFF;
                    const uint32_t* Cn = T + (has ? c : 0) * QN_WORDS;
                    const uint32_t meta = Cn[QN_META];
                    rw.logits[j] = lg4[j];
                    rw.values[j] = has ? u2f(Cn[QN_NODE_VALUE]) : 0.0f;
                    rw.rewards[j] = has ? u2f(Cn[QN_REWARD]) : 0.0f;
                    rw.discounts[j] = has ? ((meta >> 20) & 1u ? 0.0f : (t.two_player ? (1.0f * t.discount) * -1.0f : 1.0f * t.discount)) : 0.0f;
                    rw.visits[j] = has ? static_cast<int>(meta & 0xFFu) : 0;
                    rw.index[j] = has ? c : -1;
                    rw.var[j] = (has && need_var) ? u2f(Cn[QN_NODE_VAR]) : 0.0f;
                    rw.rvar[j] = 0.0f;
                }
                float cq[4];
                {
                    const float pr[4] = {pr4[0], pr4[1], pr4[2], pr4[3]};
                    qtransform_p<4, 4>(t, g, rw, pr, u2f(R[QN_RAW_VALUE]), need_var ? u2f(R[QN_RAW_VAR]) : 0.0f, beta, cq);
                }
                int lsumN = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) lsumN += rw.visits[j];
                const int sumN = gsumi<4>(lsumN);
                int act;
                if (node == 0) {
                    // gumbel_muzero_root_action_selection (A.4)
                    float gum[4];
                    uint8_t inv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { gum[j] = M.gumbel[ti][4 * q + j]; inv[j] = M.invalid[ti][4 * q + j]; }
                    const int simidx = sumN > S - 1 ? S - 1 : sumN;
                    const float cvis = static_cast<float>(t.table[static_cast<long>(M.nc[ti]) * S + simidx]);
                    act = root_pick<4, 4>(t, g, rw, cq, cvis, gum, inv);
                } else {
                    // gumbel_muzero_interior_action_selection (A.4)
                    float x[4], pr[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = rw.logits[j] + cq[j];
                    g.softmax(x, pr);
                    const float den = static_cast<float>(1 + sumN);
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = pr[j] - static_cast<float>(rw.visits[j]) / den;
                    act = g.argmax(x);
                }
                int mine = -2147483647 - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * q + j == act) mine = rw.index[j];
                const int nx = gmaxi<4>(mine);
                if (q == 0) R[QN_SEL] = (static_cast<uint32_t>(act) << 8) | (static_cast<uint32_t>(nx) & 0xFFu);
            }
            ps_wave_sync();
            EAZ_QPSTAMP(2);
            // ---- select (A.4): walk the cached choices ----
            if (tl == 0) {
                int node = 0, depth = 0, act, nx;
                while (true) {
                    const uint32_t sw = T[node * QN_WORDS + QN_SEL];
                    act = static_cast<int>((sw >> 8) & 0xFu);
                    nx = static_cast<int>(sw & 0xFFu);
                    P[depth] = static_cast<uint16_t>(node | (act << 8));
                    depth += 1;
                    if (nx == 0xFF || depth >= max_depth) break;
                    node = nx;
                }
                M.parent[ti] = node; M.action[ti] = act; M.dst[ti] = nx == 0xFF ? s : nx; M.D[ti] = depth;  // simulation s - 1 expands node s
                // place in the interpreter queue: longest expected run first -- a child's program differs from its parent's in one word, and a
                // step whose word the parent's runs never touched runs nothing (qp_env_step)
                const uint32_t ph = ST[node * QS_WORDS + QS_HDR];
                const int widx = static_cast<int>(ph & 0xFFu);  // the action becomes word step_count (of the parent)
                const bool cheap = ((ph >> 8) & 7u) != 0u || widx >= 12 || ((qp_hdr_touched(ph) >> widx) & 1u) == 0u;
                M.qkey[ti] = static_cast<uint8_t>(cheap ? 0 : 1 + qp_hdr_hint(ph));
#ifdef EAZ_FC_STAMPS
                if (step == 0 && blockIdx.x < 64) { atomicAdd(&g_qp_dhist[depth < 63 ? depth : 63], 1ull); atomicAdd(&g_qp_dhist[64 + ((s - 1) * 8) / S], static_cast<unsigned long long>(depth)); }
#endif
            }
            // the next re-score reads the prior logits / probabilities of this path's nodes: level `slot` of it, requested now
            ps_wave_sync();
            pre_node = -1;
            if (slot >= 1 && slot < M.D[ti]) {
                pre_node = P[slot] & 0xFF;
                pre_lg = *reinterpret_cast<const floatx4*>(LG + pre_node * (2 * kQpA) + 4 * q);
                pre_pr = *reinterpret_cast<const floatx4*>(LG + pre_node * (2 * kQpA) + kQpA + 4 * q);
            }
            EAZ_QPSTAMP(3);
        }
        if (it > 0) {
            // ---- E: env.step of the 16 chosen edges (s >= 1) or the real auto-reset step (s == 0), one environment per wave at a time ----
            if (tid == 0) M.env_next = 0;
            __syncthreads();
            EAZ_QPSTAMP(4);
            {
                EAZ_QP_ARGS;
                const uint32_t env_next_off = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) int*)(&M.env_next)));
                // every wave starts on its own environment, then takes what is left, one at a time: the runs are heavy-tailed
                const int n_static = 8;
                auto claim = [&]() { return n_static + lds_claim_one(env_next_off); };  // one lane, one ds_add_rtn_u32 (search_persist.hpp)
                // queue position -> environment: by descending key, ties by index (every wave ranks the 16 keys itself: lanes 0..15)
                int rank = 0;
                if (env_mode == 0) {
                    // unique keys (key, then the lower index first): rank = number of larger ones -- one compare per other environment
                    const int me = lane & 15;
                    const int kme = 16 * static_cast<int>(M.qkey[me]) + (15 - me);
                    const uint4 kq = *reinterpret_cast<const uint4*>(&M.qkey[0]);
#pragma unroll
                    for (int o = 0; o < kQpTrees; ++o) {
                        const uint32_t d = __builtin_amdgcn_readfirstlane((o >> 2) == 0 ? kq.x : ((o >> 2) == 1 ? kq.y : ((o >> 2) == 2 ? kq.z : kq.w)));
                        const int ko = 16 * static_cast<int>((d >> (8 * (o & 3))) & 0xFFu) + (15 - o);
                        rank += ko > kme ? 1 : 0;
                    }
                } else {
                    rank = lane & 15;
                }
                auto env_at = [&](int pos) {
                    if (pos >= kQpTrees) return pos;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(lane < 16 && rank == pos);
                    return static_cast<int>(__builtin_ctzll(m));
                };
                int j = env_at(wave);
                int n_done = 0;
                (void)n_done;
#pragma unroll 1
                while (j < kQpTrees) {
                    const int bj = b0 + j < B ? b0 + j : B - 1;
                    const bool jlive = b0 + j < B;
                    uint32_t* SJ = states + j * N * QS_WORDS;
                    float rew;
                    int term;
                    if (env_mode == 0) {
                        const int node = __builtin_amdgcn_readfirstlane(M.parent[j]), action = __builtin_amdgcn_readfirstlane(M.action[j]);
                        const int dstn = __builtin_amdgcn_readfirstlane(M.dst[j]);
#ifdef EAZ_FC_STAMPS
                        unsigned long long* dsp = (a.stamps && step == 0 && s == 1 + S / 2 && blockIdx.x < 256 && n_done == 0) ? a.stamps + (blockIdx.x * 8 + wave) * 16 : nullptr;
                        n_done += 1;
#else
                        unsigned long long* dsp = nullptr;
#endif
                        qp_env_step(a.e, SJ + node * QS_WORDS, SJ + dstn * QS_WORDS, M.root_ex[j], M.words[j], M.obytes[j], action, 0, env_key, bj, lane,
                                    nullptr, nullptr, rew, term, &M.task[j], a.net.l1q_mode, &M.l1var[j], dsp);
                        EAZ_QESTAMP(dsp, 14);
#ifdef EAZ_FC_STAMPS
                        if (step == 0 && blockIdx.x < 64 && s <= 64) {
                            ps_wave_sync();
                            if (lane == 0)
                                g_qp_its[(blockIdx.x * 64 + (s - 1)) * 16 + j] = (4u * qp_hdr_hint(SJ[dstn * QS_WORDS + QS_HDR])) | ((4u * qp_hdr_hint(SJ[node * QS_WORDS + QS_HDR])) << 8) | 0x10000u;
                        }
#endif
                        if (lane == 0) { M.o_reward[j] = rew; M.o_term[j] = term; }  // rewards[b, current_player], single player
                    } else {
                        // vmap(auto_reset(env.step, env.init)) (selfplay.py:26-75,135) on the environment's current state
                        const int action = __builtin_amdgcn_readfirstlane(M.real_action[j]);
                        const long o = static_cast<long>(step - 1) * B + bj;
                        char* blob0 = (jlive && a.sp.out_states) ? a.sp.out_states + o * sizeof(eaz_subleq_state) : nullptr;
                        // the caller's state batch gets the final states; evaluate: after every step (which one is the last is not known ahead)
                        char* blob1 = (jlive && (step == n_steps || evaluate)) ? a.sp.states + static_cast<long>(bj) * sizeof(eaz_subleq_state) : nullptr;
                        qp_env_step(a.e, M.root_state[j], M.root_state[j], M.root_ex[j], M.words[j], M.obytes[j], action, evaluate ? 2 : 1, env_key, bj, lane,
                                    blob0, blob1, rew, term, &M.task[j], a.net.l1q_mode, &M.l1var[j]);
                        if (evaluate && lane == 0) {  // evaluate.py:47-50: plain env.step, rewards[b, current_player] summed (single player)
                            M.sum_reward[j] = M.sum_reward[j] + rew;
                            M.real_term[j] = static_cast<uint8_t>(term);
                        }
                    }
                    j = env_at(claim());
                }
            }
            EAZ_QPSTAMP(5);
            __syncthreads();
            if (evaluate && env_mode != 0 && all_terminated()) { steps_done = step; break; }  // cond_fn, evaluate.py:18-20 (the cap: it == n_iter)
            if (it == n_iter) break;
        }
        if (s == 0) {
            // ---- a new real step: empty tree (mctx instantiate_tree_from_root, A.3) ----
            EAZ_QP_TREE_LANES;
            for (int n = tl; n < N; n += 32) {
                uint32_t* R = T + n * QN_WORDS;
#pragma unroll
                for (int w = 0; w < QN_WORDS; ++w) R[w] = w >= QN_SEL ? 0xFFFFFFFFu : 0u;
            }
        }
        // ---- F: the forward of the three heads for the 16 observations whose row-words are in M.words ----
        EAZ_QPSTAMP(6);
        if (wave < 4) {
            // ================= waves 0-3: layers 1-3 of the three heads for the 16 rows; barriers: H1 of head 0, 1, 2 | quarter sums =================
            EAZ_QP_ARGS;
            EAZ_QP_FWD_LANES;
            const int n1 = a.net.l1q_halves;  // half-rounds of layer 1: 7 (224 packed inputs) or 5 (144 + 16 zero rows), sq16_l1_halves
            const int l1_mode = a.net.l1q_mode;
            // layer-1 mode 1: where the chains of this lane's four rows start (the task's vector, or the root blob's for a state that carries its rows)
            const float* l1src[4];
            floatx4 accn[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 4 * fkq + reg;
                const int task = M.task[row];
                const int branch = task - 1 < 0 ? 0 : (task - 1 > 5 ? 5 : task - 1);
                l1src[reg] = (M.l1var[row] ? a.l1init + kQpL1Tab + static_cast<long>(b0 + row) * (3 * kFcH) : a.l1init + branch * (3 * kFcH)) + fcol;
                accn[reg] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                if (l1_mode == 1) accn[reg] = *reinterpret_cast<const floatx4*>(l1src[reg]);  // head 0
            }
            __amdgpu_buffer_rsrc_t W1r[3], W2r[3];
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                W1r[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.net.w1q[a.heads.id[h]]), 0, n1 * kQpHalf * 4096, 0x00020000);
                W2r[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.net.w[a.heads.id[h] * 3 + 1]), 0, kFcH * kFcH * 4, 0x00020000);
            }
            // layer-3 weights of this wave's quarter of k as MFMA B operands (W3[64 w + 4 q + kq][r]): ONE set of 16 registers, requested
            // ~10 k cycles before the chain that consumes it.  First the two one-output heads together: B column 0 = W3 of the value head,
            // column 1 = W3 of the ube head (a chain then also produces the other head's column from the wrong hidden tile, which nobody
            // reads); once the ube head's chain has run (first half-rounds of the third head) the policy head's weights take their place.
            float w3v[16];
            {
                const float* pwv = a.net.w[a.heads.id[fr == 1 ? 1 : 0] * 3 + 2] + (fn0 + fkq);
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) { const float wv = pwv[4 * qq]; w3v[qq] = fr < 2 ? wv : 0.0f; }
            }
            // this lane's layer-1 A operands: half-round j = the 8 FP8 bytes 64 kq + 8 j .. + 7 of the row (16-byte words swizzled, qp_ob_swz)
            const uint8_t* xrow_base = &M.obytes[fr][0];
            const int xswz = qp_ob_swz(xrow_base);
            auto xrow = [&](int j) { return *reinterpret_cast<const uint2*>(xrow_base + ((((4 * fkq + (j >> 1)) ^ xswz) << 4) | ((j & 1) << 3))); };
            const float* hpA = H2 + h16_off(fn0 + fkq, fr);  // quarter chain: H2[h16_off(n0 + 4 q + kq, r)] = hpA[68 q]
            float av[kPsARing];
            constexpr int kHalfBytes = kQpHalf * 4096;  // one half-round of the weight stream
            EAZ_QWSTAMP(0);
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const int P = h & 1;               // phase of this head's first half-round (qp_half)
                const int hn = h == 2 ? 0 : h + 1;  // the head whose layer 1 follows in the stream
                floatx4 acc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                if (l1_mode == 1) {
                    // the chains of row 4 kq + reg start from the partial sum over the row's constant inputs (qp_l1_chain), requested a head ago
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c][reg] = accn[reg][c];
                    }
                }
                floatx4 a3 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                // ---- layer 1: [16 x 32 n1] x [32 n1 x 256] on the packed inputs (sq16_l1_src); half-round e refills its registers for half-round
                // e + 2 of the head's sequence L1_0 .. L1_(n1-1), L2_0 .. L2_7; the first two carry the quarter chain of the head before ----
                uint2 xcur = xrow(0), xnext = xrow(1);
                {
                    const floatx2 f = qp_abytes2(xcur, 0);
                    av[0] = f[0]; av[1] = f[1];
                }
                if (h > 0) qp_half_ph<1, true>(P, acc, Bq, av, nullptr, xcur, xnext, W1r[h], fvoff, 2 * kHalfBytes, a3, hpA, w3v);
                else qp_half_ph<0, true>(P, acc, Bq, av, nullptr, xcur, xnext, W1r[h], fvoff, 2 * kHalfBytes, a3, hpA, w3v);
                xcur = xnext; xnext = xrow(2);
                if (h > 0) qp_half_ph<2, true>(P ^ 1, acc, Bq, av, nullptr, xcur, xnext, W1r[h], fvoff, 3 * kHalfBytes, a3, hpA, w3v);
                else qp_half_ph<0, true>(P ^ 1, acc, Bq, av, nullptr, xcur, xnext, W1r[h], fvoff, 3 * kHalfBytes, a3, hpA, w3v);
                EAZ_QRSTAMP(h * 16 + 1);
                if (h > 0 && fr == h - 1) {  // head 0 = value: output column 0, head 1 = ube: column 1 (shared W3 registers)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) part[(fw * 16 + 4 * fkq + reg) * kQpPartLd + fr] = a3[reg];
                }
                if (h == 2) {
                    const float* pwp = a.net.w[a.heads.id[2] * 3 + 2] + (fn0 + fkq) * kQpA + fr;
#pragma unroll
                    for (int qq = 0; qq < 16; ++qq) w3v[qq] = pwp[4 * qq * kQpA];
                }
                // n1 is odd: pairs (2, 3), (4, 5), ... then the single half-round n1 - 1.  The last pair is peeled: inside the loop every refill is
                // the same head's W1 at a running offset (scalar work per iteration: two adds; the selects of a generic body sat outside the
                // MFMAs' shadow behind the sched_barriers and cost ~100 clocks per iteration)
#pragma unroll 1
                for (int j = 2; j + 3 < n1; j += 2) {
                    xcur = xnext; xnext = xrow(j + 1);
                    qp_half_ph<0, true>(P, acc, Bq, av, nullptr, xcur, xnext, W1r[h], fvoff, (j + 2) * kHalfBytes, a3, hpA, w3v);
                    xcur = xnext; xnext = xrow(j + 2);
                    qp_half_ph<0, true>(P ^ 1, acc, Bq, av, nullptr, xcur, xnext, W1r[h], fvoff, (j + 3) * kHalfBytes, a3, hpA, w3v);
                    EAZ_QRSTAMP(h * 16 + j + 1);
                }
                xcur = xnext; xnext = xrow(n1 - 2);
                qp_half_ph<0, true>(P, acc, Bq, av, nullptr, xcur, xnext, W1r[h], fvoff, (n1 - 1) * kHalfBytes, a3, hpA, w3v);  // half-round n1 - 3 -> n1 - 1
                xcur = xnext; xnext = xrow(n1 - 1);
                qp_half_ph<0, true>(P ^ 1, acc, Bq, av, nullptr, xcur, xnext, W2r[h], fvoff, 0, a3, hpA, w3v);                   // n1 - 2 -> layer 2, half-round 0
                EAZ_QRSTAMP(h * 16 + n1 - 2);
                xcur = xnext;  // (the last half-round converts nothing of `abits_next` that is used: layer 2 refills the ring)
                qp_half_ph<0, true>(P, acc, Bq, av, nullptr, xcur, xcur, W2r[h], fvoff, kHalfBytes, a3, hpA, w3v);               // n1 - 1 -> layer 2, half-round 1
                EAZ_QRSTAMP(h * 16 + n1 - 1);
                EAZ_QWSTAMP(1 + 2 * h);
                {
                    float* Ht = H1 + (h & 1) * kF16HFloats;
                    const floatx4 bb1 = *reinterpret_cast<const floatx4*>(&M.b12[h][0][fcol]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        floatx4 o;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const float v = acc[c][reg] + bb1[c];
                            o[reg] = v > 0.0f ? v : 0.0f;
                        }
                        *reinterpret_cast<floatx4*>(Ht + h16_off(fcol + c, 4 * fkq)) = o;
                    }
                }
                __syncthreads();  // H1 of head h
                if (l1_mode == 1 && h < 2) {  // the next head's chain starts: in flight during layer 2
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) accn[reg] = *reinterpret_cast<const floatx4*>(l1src[reg] + (h + 1) * kFcH);
                }
                // ---- layer 2: [16 x 256] x [256 x 256], half-rounds in phases P ^ 1, P, ...; the refills of the last two are the first two
                // half-rounds of the next head (third head: of the next forward, whose first phase is 0 again -- the skipped phase) ----
                floatx4 acc2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc2[c] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                const float* Ap = H1 + (h & 1) * kF16HFloats + fkq * 16 + fr;  // h16_off(4 ks + kq, r) = 68 ks + 16 kq + r
#pragma unroll
                for (int i = 0; i < kPsARing; ++i) av[i] = Ap[i * 68];
                // reads past k-step 63 land in the next tile and are never used
#pragma unroll 1
                for (int j = 0; j < 6; j += 2) {
                    qp_half_ph<0, false>(P ^ 1, acc2, Bq, av, Ap + (j * kQpHalf + kPsARing) * 68, uint2{}, uint2{}, W2r[h], fvoff, (j + 2) * kHalfBytes, a3, hpA, w3v);
                    qp_half_ph<0, false>(P, acc2, Bq, av, Ap + ((j + 1) * kQpHalf + kPsARing) * 68, uint2{}, uint2{}, W2r[h], fvoff, (j + 3) * kHalfBytes, a3, hpA, w3v);
                    EAZ_QRSTAMP(h * 16 + 9 + j);
                }
                // the last pair refills for the first two half-rounds of the next head; after the third head the next forward starts in
                // phase 0 again (one phase is skipped), so its two targets swap
                qp_half_ph<0, false>(P ^ 1, acc2, Bq, av, Ap + (6 * kQpHalf + kPsARing) * 68, uint2{}, uint2{}, W1r[hn], fvoff, h == 2 ? kHalfBytes : 0, a3, hpA, w3v);
                qp_half_ph<0, false>(P, acc2, Bq, av, Ap + (7 * kQpHalf + kPsARing) * 68, uint2{}, uint2{}, W1r[hn], fvoff, h == 2 ? 0 : kHalfBytes, a3, hpA, w3v);
                EAZ_QRSTAMP(h * 16 + 15);
                EAZ_QWSTAMP(2 + 2 * h);
                const floatx4 bb2 = *reinterpret_cast<const floatx4*>(&M.b12[h][1][fcol]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    floatx4 o;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float v = acc2[c][reg] + bb2[c];
                        o[reg] = v > 0.0f ? v : 0.0f;
                    }
                    *reinterpret_cast<floatx4*>(H2 + h16_off(fcol + c, 4 * fkq)) = o;
                }
                if (h == 2) {
                    // the last head's quarter chain has no layer 1 to hide under: 16 dependent MFMAs (k ascending = the oracle's fmaf chain)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    floatx4 z3 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int qq = 0; qq < 16; ++qq) z3 = __builtin_amdgcn_mfma_f32_16x16x4f32(hpA[68 * qq], w3v[qq], z3, 0, 0, 0);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) part[(fw * 16 + 4 * fkq + reg) * kQpPartLd + 2 + fr] = z3[reg];
                }
            }
        } else {
            // ================= waves 4-7: the same barriers; wave 4 looks the 16 observations' novelty up =================
            EAZ_QP_ARGS;
            if (wave == 4) qp_novelty(a.net, M, lane);
            __syncthreads(); __syncthreads(); __syncthreads();  // H1 of head 0, 1, 2
        }
        EAZ_QPSTAMP(7);
        __syncthreads();  // the quarter sums of the three heads
        if (s == 0) {
            // ---- root of real step `step`: selfplay.py:92-106,113 (root_prep_kernel) + mctx instantiate_tree_from_root (tree_root_kernel) ----
            EAZ_QP_ARGS;
            EAZ_QP_TREE_LANES;
            const Key kt = fold(a.sp.key, kTagStep + static_cast<uint32_t>(step));
            const Key gkey = evaluate ? Key{0u, 0u} : fold(kt, kTagGumbel);  // evaluate: gumbel_scale = 0 (evaluate.py:44), the noise of the host driver's key
            const float z = tl < 18 ? head_out(ti, tl < 16 ? 2 + tl : tl - 16) : 0.0f;
            const float th = (tl == 16 || tl == 17) ? det_tanh(z) : 0.0f;
            if (tl == 16) M.o_value[ti] = th;                                         // fully_connected.py:55
            if (tl == 17) M.o_ube[ti] = (0.5f * (th + 1.0f)) * a.net.max_u;           // :64, :93
            ps_wave_sync();
            float lgv = -kInf;
            if (tl < 16) {
                lgv = (a.sp.flags & kSpUniform) ? 1.0f : z;                            // :96-98
                const long gi = static_cast<long>(gb) * kQpA + tl;
                const float gn = a.sp.gumbel ? a.sp.gumbel[static_cast<long>(step) * B * kQpA + gi] : gumbel(gkey, static_cast<uint32_t>(gi));
                M.gumbel[ti][tl] = a.sp.gumbel_scale * gn;
            }
            const float m = gmax<16>(lgv);  // mask_invalid_actions (A.2): every action legal (subleq.py:626)
            {
                const float ml = lgv - m;
                const float pm = gmax<16>(ml);
                const float pe = tl < 16 ? det_exp(ml - pm) : 0.0f;
                const float ps = gsum<16>(pe);
                if (tl < 16) { M.root_logits[ti][tl] = ml; M.root_probs[ti][tl] = pe / ps; }
            }
            if (tl == 0) {
                const float root_v = M.o_value[ti];
                const float root_u = ube_combine(M.o_ube[ti], M.o_nov[ti], t.max_u);  // fully_connected.py:92-96
                T[QN_RAW_VALUE] = f2u(root_v); T[QN_NODE_VALUE] = f2u(root_v);
                T[QN_RAW_VAR] = f2u(root_u); T[QN_NODE_VAR] = f2u(root_u);
                T[QN_META] = 1u | (0xFFu << 8);
                M.root_v[ti] = root_v; M.root_u[ti] = root_u;
                M.D[ti] = 0; M.parent[ti] = 0; M.action[ti] = 0; M.dst[ti] = 0;
            }
            if (tl < QS_WORDS) ST[tl] = M.root_state[ti][tl];
            ps_wave_sync();
        }
    }
    if (evaluate) {
        EAZ_QP_ARGS;
        if (tid < kQpTrees && b0 + tid < B) a.sp.eval_sum[b0 + tid] = M.sum_reward[tid];
        if (tid == 0) { a.sp.eval_wg_iters[blockIdx.x] = steps_done; atomicMax(a.sp.eval_iters, steps_done); }
        return;
    }
    if (!search_only) return;  // self-play never reads the tree arrays
    // ---------------------------------------------------------------- search only: the trees back to the mctx arrays -----------------------
    // Every field of every node (the caller's eaz_search_begin skipped the clear: unexpanded nodes get mctx's initial values, ints -1 / 0,
    // floats 0); children_X[n, a] = the child's node field, the edge's reward / discount from the child's record (header of this file).
    {
        EAZ_QP_ARGS;
        EAZ_QP_TREE_LANES;
        ps_wave_sync();
        if (tree_live) {
            const long nb = static_cast<long>(gb) * N;
            const float disc_live = t.two_player ? (1.0f * t.discount) * -1.0f : 1.0f * t.discount;
            for (int n = tl; n < N; n += 32) {
                const uint32_t* R = T + n * QN_WORDS;
                const uint32_t meta = R[QN_META];
                const int visits = static_cast<int>(meta & 0xFFu);
                t.raw_values[nb + n] = u2f(R[QN_RAW_VALUE]); t.node_values[nb + n] = u2f(R[QN_NODE_VALUE]);
                t.raw_var[nb + n] = u2f(R[QN_RAW_VAR]); t.node_var[nb + n] = u2f(R[QN_NODE_VAR]);
                t.node_visits[nb + n] = visits;
                const bool linked = n > 0 && visits > 0;
                t.parents[nb + n] = linked ? static_cast<int>((meta >> 8) & 0xFFu) : -1;
                t.action_from_parent[nb + n] = linked ? static_cast<int>((meta >> 16) & 0xFu) : -1;
            }
            for (int i = tl; i < N * kQpA; i += 32) {
                const int n = i >> 4, act = i & 15;
                const uint32_t* R = T + n * QN_WORDS;
                const int c = (R[QN_CHILD + (act >> 2)] >> (8 * (act & 3))) & 0xFFu;
                const bool has = c != 0xFF;
                const uint32_t* Cn = T + (has ? c : 0) * QN_WORDS;
                const uint32_t cm = Cn[QN_META];
                const long o = (nb + n) * kQpA + act;
                t.children_index[o] = has ? c : -1;
                t.children_visits[o] = has ? static_cast<int>(cm & 0xFFu) : 0;
                t.children_values[o] = has ? u2f(Cn[QN_NODE_VALUE]) : 0.0f;
                t.children_var[o] = has ? u2f(Cn[QN_NODE_VAR]) : 0.0f;
                t.children_rewards[o] = has ? u2f(Cn[QN_REWARD]) : 0.0f;
                t.children_discounts[o] = has ? ((cm >> 20) & 1u ? 0.0f : disc_live) : 0.0f;
                t.children_reward_var[o] = 0.0f;
                const bool expanded = (R[QN_META] & 0xFFu) != 0;
                t.children_prior_logits[o] = n == 0 ? M.root_logits[ti][act] : (expanded ? LG[n * (2 * kQpA) + act] : 0.0f);
            }
            // embeddings of the expanded nodes: compact state -> the 336-byte blob of include/eaz.h (node 0 is the caller's)
            for (int n = 1 + (tl >> 4); n < N; n += 2) {
                const int l16 = tl & 15;
                const uint32_t* R = T + n * QN_WORDS;
                if ((R[QN_META] & 0xFFu) == 0) continue;
                const uint32_t* sw = ST + n * QS_WORDS;
                const unsigned long long mem = qp_u64(sw[QS_MEM], sw[QS_MEM + 1]), inaf = qp_u64(sw[QS_INAF], sw[QS_INAF + 1]);
                const unsigned long long outaf = qp_u64(sw[QS_OUTAF], sw[QS_OUTAF + 1]);
                const uint32_t hdr = sw[QS_HDR];
                const int task = M.task[ti];
                const int branch = task - 1 < 0 ? 0 : (task - 1 > 5 ? 5 : task - 1);
                const bool exroot = ((hdr >> 11) & 1u) != 0;
                const unsigned long long exin = exroot ? qp_u64(M.root_ex[ti][0], M.root_ex[ti][1]) : kSq16Cases.v[branch][0][0];
                const unsigned long long exout = exroot ? qp_u64(M.root_ex[ti][2], M.root_ex[ti][3]) : kSq16Cases.v[branch][0][1];
                char* blob = t.embeddings + (nb + n) * sizeof(eaz_subleq_state);
                if (l16 == 0)
                    *reinterpret_cast<uint4*>(blob) = make_uint4(hdr & 0xFFu, static_cast<uint32_t>(task), sw[QS_REWARD],
                                                                 ((hdr >> 8) & 1u) | (((hdr >> 9) & 1u) << 8) | (((hdr >> 10) & 1u) << 16));
                // rows: 32 uint16 = lane l16 writes elements 2 l16 and 2 l16 + 1
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int el = 2 * l16 + e, row = el >> 3, sh = 8 * (el & 7);
                    const unsigned long long srcv = row == 0 ? exin : (row == 1 ? inaf : (row == 2 ? exout : outaf));
                    reinterpret_cast<uint16_t*>(blob + 16)[el] = static_cast<uint16_t>((srcv >> sh) & 0xFFull);
                }
                uint4 cbytes = make_uint4(0u, 0u, 0u, 0u);
                if (l16 == 0) {
                    uint32_t w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t n4 = static_cast<uint32_t>(mem >> (16 * i)) & 0xFFFFu;
                        w[i] = (n4 & 0xFu) | ((n4 & 0xF0u) << 4) | ((n4 & 0xF00u) << 8) | ((n4 & 0xF000u) << 12);
                    }
                    cbytes = make_uint4(w[0], w[1], w[2], w[3]);
                }
                *reinterpret_cast<uint4*>(blob + 80 + 16 * l16) = cbytes;
            }
        }
    }
}

}  // namespace eaz
