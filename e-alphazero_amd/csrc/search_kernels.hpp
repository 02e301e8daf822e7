// search_kernels.hpp -- the kernels around the per-simulation tree kernel: tree instantiation, root preparation,
// final action / action_weights, epistemic_summary, action choice of the self-play driver.
//
// Reference: mctx instantiate_tree_from_root / gumbel_muzero_policy tail / Tree.summary [EXT]
// and src/selfplay.py:92-134.
#pragma once
#include "env_kernels.hpp"
#include "rng.hpp"
#include "tree.hpp"

namespace eaz {

// mctx instantiate_tree_from_root (A.3): ints -1 / 0, floats 0.  Grid-stride over B*N*A.
// keep_root: the root node (written by tree_root_kernel before this launch) is left alone -- the deferred clear of a search whose
// eaz_search_begin skipped it (eaz_api.hip ensure_cleared).
__global__ void __launch_bounds__(256) tree_clear_kernel(TreeDev t, int keep_root) {
    const long bn = static_cast<long>(t.B) * t.N, bna = bn * t.A;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    const bool rows = t.lazy_rows == 0;  // lazy rows (tree.hpp): the [B,N,A] arrays are left as they are
    for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < (rows ? bna : bn); i += stride) {
        // i indexes the [B,N,A] arrays, and (while i < B*N) the [B,N] arrays as well
        if (rows) {
            const bool root_row = keep_root && (i / t.A) % t.N == 0;
            t.children_index[i] = -1;
            t.children_visits[i] = 0;
            if (!root_row) t.children_prior_logits[i] = 0.0f;
            t.children_values[i] = 0.0f;
            t.children_rewards[i] = 0.0f;
            t.children_discounts[i] = 0.0f;
            t.children_var[i] = 0.0f;
            t.children_reward_var[i] = 0.0f;
        }
        if (i < bn) {
            t.parents[i] = -1;
            t.action_from_parent[i] = -1;
            if (!(keep_root && i % t.N == 0)) {
                t.node_visits[i] = 0;
                t.raw_values[i] = 0.0f;
                t.node_values[i] = 0.0f;
                t.raw_var[i] = 0.0f;
                t.node_var[i] = 0.0f;
            }
        }
    }
}

// Root write: mask_invalid_actions(prior_logits) (A.2), update_tree_node(ROOT) (A.3), embedding[ROOT] = root state.
template <int LPT, int APL>
__global__ void __launch_bounds__(256) tree_root_kernel(TreeDev t, const float* prior_logits, const float* value,
                                                        const float* value_var, const char* root_states, const float* beta,
                                                        const uint8_t* invalid, const float* gumbel) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gtid / LPT;
    const bool live = b < t.B;
    const Group<LPT, APL> g = make_group<LPT, APL>(live ? b : 0, gtid % LPT, t.A);
    const int A = t.A;
    const long nb = static_cast<long>(g.b) * t.N;
    float lg[APL];
    bool inv[APL];
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        const long o = static_cast<long>(g.b) * A + g.act[j];
        lg[j] = g.ok[j] ? prior_logits[o] : 0.0f;
        inv[j] = g.ok[j] && invalid && invalid[o];
    }
    const float m = g.vmax(lg);
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        if (!(live && g.ok[j])) continue;
        const long o = static_cast<long>(g.b) * A + g.act[j];
        t.children_prior_logits[nb * A + g.act[j]] = inv[j] ? kF32Min : lg[j] - m;
        if (t.lazy_rows) t.children_visits[nb * A + g.act[j]] = 0;  // the root's row of the one array lazy rows keep initialised (tree.hpp)
        t.root_invalid[o] = inv[j] ? 1 : 0;
        t.root_gumbel[o] = gumbel[o];
    }
    if (live && g.gl == 0) {
        t.beta[g.b] = beta[g.b];
        t.raw_values[nb] = value[g.b];
        t.node_values[nb] = value[g.b];
        t.raw_var[nb] = value_var[g.b];
        t.node_var[nb] = value_var[g.b];
        t.node_visits[nb] = 1;
    }
    // embedding copy: state_bytes is a multiple of 16
    if (live) {
        const uint4* src = reinterpret_cast<const uint4*>(root_states + static_cast<long>(g.b) * t.state_bytes);
        uint4* dst = reinterpret_cast<uint4*>(t.embeddings + nb * t.state_bytes);
        for (int i = g.gl; i < t.state_bytes / 16; i += LPT) dst[i] = src[i];
    }
}

// Tail of gumbel_muzero_policy (A.2): action = masked_argmax(score_considered(max visit, ...)),
// action_weights = softmax(mask(root_logits + completed_q)).
template <int LPT, int APL>
__global__ void __launch_bounds__(256) tree_finish_kernel(TreeDev t, int32_t* action_out, float* action_weights) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gtid / LPT;
    const bool live = b < t.B;
    const Group<LPT, APL> g = make_group<LPT, APL>(live ? b : 0, gtid % LPT, t.A);
    const int A = t.A;
    const long nb = static_cast<long>(g.b) * t.N;
    const float beta = t.beta[g.b];
    NodeRows<APL> r;
    load_rows<LPT, APL>(t, g, 0, beta != 0.0f, r);
    float cq[APL];
    qtransform<LPT, APL>(t, g, r, t.raw_values[nb], t.raw_var[nb], beta, cq);
    int lmax = 0;
#pragma unroll
    for (int j = 0; j < APL; ++j) lmax = r.visits[j] > lmax ? r.visits[j] : lmax;
    const int maxN = gmaxi<LPT>(lmax);
    float r_gumbel[APL];
    uint8_t r_invalid[APL];
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        const long o = static_cast<long>(g.b) * A + (g.ok[j] ? g.act[j] : 0);
        r_gumbel[j] = t.root_gumbel[o];
        r_invalid[j] = t.root_invalid[o];
    }
    const int action = root_pick<LPT, APL>(t, g, r, cq, static_cast<float>(maxN), r_gumbel, r_invalid);
    float x[APL], p[APL];
#pragma unroll
    for (int j = 0; j < APL; ++j) x[j] = r.logits[j] + cq[j];
    const float m = g.vmax(x);
#pragma unroll
    for (int j = 0; j < APL; ++j)
        if (g.ok[j]) x[j] = t.root_invalid[static_cast<long>(g.b) * A + g.act[j]] ? kF32Min : x[j] - m;
    g.softmax(x, p);
    if (!live) return;
    if (g.gl == 0 && action_out) action_out[g.b] = action;
#pragma unroll
    for (int j = 0; j < APL; ++j)
        if (g.ok[j] && action_weights) action_weights[static_cast<long>(g.b) * A + g.act[j]] = p[j];
}

// emctx Tree.epistemic_summary() (A.7): one thread per (b, a); visit totals are exact small integers.
__global__ void __launch_bounds__(256) tree_summary_kernel(TreeDev t, float* value, float* value_std, float* qvalues,
                                                           float* qvalues_var, float* visit_counts, float* visit_probs) {
    const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<long>(t.B) * t.A) return;
    const int b = static_cast<int>(i / t.A), a = static_cast<int>(i - static_cast<long>(b) * t.A);
    const long nb = static_cast<long>(b) * t.N, o = nb * t.A;
    if (a == 0) {
        if (value) value[b] = t.node_values[nb];
        if (value_std) value_std[b] = __builtin_sqrtf(t.node_var[nb]);
    }
    // sum of the root's children_visits: every backward counts the root and exactly one of its edges (A.6), and the root starts at 1 (A.3),
    // so the sum is node_visits[root] - 1 (round 3 added the A counts up per thread: 65 536 loads per tree at 256 actions, 68 us per call)
    const int tot = t.node_visits[nb] - 1;
    const int vis = t.children_visits[o + a];
    const bool linked = !t.lazy_rows || vis > 0;  // lazy rows (tree.hpp): an edge never taken reads as mctx's initial values
    const float d = linked ? t.children_discounts[o + a] : 0.0f;
    if (qvalues) qvalues[i] = (linked ? t.children_rewards[o + a] : 0.0f) + d * (linked ? t.children_values[o + a] : 0.0f);
    if (qvalues_var) qvalues_var[i] = (linked ? t.children_reward_var[o + a] : 0.0f) + (d * d) * (linked ? t.children_var[o + a] : 0.0f);
    const float vc = static_cast<float>(vis);
    const float ftot = static_cast<float>(tot);
    if (visit_counts) visit_counts[i] = vc;
    // mctx Tree.summary [EXT]: visit_counts / max(total, 1) -- all zeros for a root that has not been searched
    if (visit_probs) visit_probs[i] = vc / (ftot > 1.0f ? ftot : 1.0f);
}

// Root preparation of selfplay's step_fn (src/selfplay.py:92-106,113): policy logits (uniform option), per-env beta
// schedule beta * linspace(0, 1, B), invalid = ~legal_action_mask, root Gumbel noise (explicit or generated).
__global__ void __launch_bounds__(256) root_prep_kernel(int B, int A, const float* logits_in, int uniform_policy, float sbeta,
                                                        const uint8_t* legal, const float* gumbel_in, const Key* step_keys,
                                                        float gumbel_scale, float* logits_out, float* beta, uint8_t* invalid,
                                                        float* gumbel_out, float* root_ube, const float* novelty, float max_u,
                                                        int constant_beta) {
    const Key gkey = step_keys[0];
    const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<long>(B) * A) return;
    const int b = static_cast<int>(i / A);
    logits_out[i] = uniform_policy ? 1.0f : logits_in[i];                                   // :96-98
    invalid[i] = legal ? (legal[i] ? 0 : 1) : 0;                                            // :113
    const float g = gumbel_in ? gumbel_in[i] : gumbel(gkey, static_cast<uint32_t>(i));
    gumbel_out[i] = gumbel_scale * g;
    if (i == static_cast<long>(b) * A) {
        // selfplay.py:105 beta * linspace(0, 1, B); evaluate.py:34 / reanalyze.py:75: beta * ones
        beta[b] = constant_beta ? sbeta * 1.0f : sbeta * (B > 1 ? static_cast<float>(b) / static_cast<float>(B - 1) : 0.0f);
        root_ube[b] = ube_combine(root_ube[b], novelty[b], max_u);  // fully_connected.py:92-96 joined with the novelty lookup
    }
}

// Action choice (src/selfplay.py:118-134): search action, or categorical over visit PROBABILITIES / action weights
// used as logits (reproduced literally), categorical(key, logits) = argmax(logits + Gumbel).
__global__ void __launch_bounds__(256) choose_action_kernel(int B, int A, int32_t* action, const float* visit_probs,
                                                            const float* action_weights, int sample_actions,
                                                            int sample_improved, const Key* step_keys) {
    const Key kvis = step_keys[2], kimp = step_keys[3];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int act = action[b];
    if (sample_actions) {
        float best = -kInf; int bi = 0;
        for (int a = 0; a < A; ++a) {
            const float v = visit_probs[static_cast<long>(b) * A + a] + gumbel(kvis, static_cast<uint32_t>(b * A + a));
            if (v > best) { best = v; bi = a; }
        }
        act = bi;
    }
    if (sample_improved) {
        float best = -kInf; int bi = 0;
        for (int a = 0; a < A; ++a) {
            const float v = action_weights[static_cast<long>(b) * A + a] + gumbel(kimp, static_cast<uint32_t>(b * A + a));
            if (v > best) { best = v; bi = a; }
        }
        act = bi;
    }
    action[b] = act;
}

// uniformrandomplay's actions (src/selfplay.py:160): randint(0, num_actions)
__global__ void __launch_bounds__(256) random_action_kernel(int B, int A, Key k, int32_t* action) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) action[b] = static_cast<int32_t>(rbits(k, static_cast<uint32_t>(b)) % static_cast<uint32_t>(A));
}

// evaluate.py:18-20,48-50: number of not-yet-terminated envs + return accumulation
__global__ void __launch_bounds__(256) count_alive_kernel(const uint8_t* terminated, int B, int* alive) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && !terminated[b]) atomicAdd(alive, 1);
}
__global__ void __launch_bounds__(256) accumulate_kernel(float* sum, const float* reward, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) sum[b] = sum[b] + reward[b];
}

// reanalyze.py:88-122 target construction, one group per batch element (same reductions as the tree kernel)
template <int LPT, int APL>
__global__ void __launch_bounds__(256) reanalyze_targets_kernel(int B, int A, const int32_t* action, const float* qvalues, const float* qvar,
                                                                const float* visit_counts, const float* root_value, const float* root_std,
                                                                const uint8_t* invalid, const float* next_value, const float* next_reward,
                                                                const uint8_t* next_term, const uint8_t* term, float discount,
                                                                float exploration_beta, int exploration_ube_target, float temperature,
                                                                float* value_target, float* ube_target, float* exploration_policy_target) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gtid / LPT;
    const bool live = b < B;
    const Group<LPT, APL> g = make_group<LPT, APL>(live ? b : 0, gtid % LPT, A);
    const int act = action[g.b];
    float q[APL], qv[APL], c[APL], p[APL];
    const float vfill = root_value[g.b] + exploration_beta * root_std[g.b];                                      // :117
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        const long o = static_cast<long>(g.b) * A + g.act[j];
        q[j] = g.ok[j] ? qvalues[o] : 0.0f;
        qv[j] = g.ok[j] ? qvar[o] : 0.0f;
        const float sc = q[j] + exploration_beta * __builtin_sqrtf(qv[j]);                                        // :115
        c[j] = (g.ok[j] && visit_counts[o] > 0.0f) ? sc : vfill;                                                  // complete_qs :32-41
    }
    const float mx = g.vmax(qv);                                                                                  // :103
    float q_act = -kInf, qv_act = -kInf;
#pragma unroll
    for (int j = 0; j < APL; ++j)
        if (g.ok[j] && g.act[j] == act) { q_act = q[j]; qv_act = qv[j]; }
    q_act = gmax<LPT>(q_act);                                                                                     // :88
    qv_act = gmax<LPT>(qv_act);                                                                                   // :104
    const float m = g.vmax(c);                                                                                    // mask_invalid_actions :16-29
#pragma unroll
    for (int j = 0; j < APL; ++j)
        if (g.ok[j]) c[j] = (invalid[static_cast<long>(g.b) * A + g.act[j]] ? kF32Min : c[j] - m) * temperature;    // :120-122
    g.softmax(c, p);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < APL; ++j)
        if (g.ok[j]) exploration_policy_target[static_cast<long>(g.b) * A + g.act[j]] = p[j];
    if (g.gl == 0) {
        const float not_term1 = next_term[g.b] ? 0.0f : 1.0f, not_term0 = term[g.b] ? 0.0f : 1.0f;
        const float from_td = next_reward[g.b] + (discount * next_value[g.b]) * not_term1;                        // :95-96
        const float vt = q_act > from_td ? q_act : from_td;                                                       // :102
        const float ut = exploration_ube_target ? mx : qv_act;                                                    // :106
        value_target[g.b] = vt * not_term0;                                                                       // :109
        ube_target[g.b] = ut * not_term0;                                                                         // :110
    }
}

// ---- Subleq word_size 16: tree step + leaf step in one launch ---------------------------------------------------------------------
// One wavefront per tree: [expand + backward of simulation sim-1, select of simulation sim] on its first 16 lanes (one per action;
// the other lanes mirror them), then env.step of the chosen edge by the one-wavefront interpreter (subleq16_wave_body): embedding
// slot of the parent -> slot of the child, plus the child's observation row and novelty for the network forward.  Same arithmetic
// as tree_step_kernel<16, 1> followed by subleq16_wave_kernel; what goes is one dependent launch per simulation (a graph edge
// costs about as much as the select walk) and the round trip of (parent, action, child slot) through memory.
__global__ void __launch_bounds__(64) subleq16_tree_leaf_kernel(TreeDev t, int sim, int mode, EnvDev e, float* next_reward, uint8_t* next_term,
                                                                uint8_t* next_legal, uint8_t* obs_out, NoveltyFuse nf) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const TreeSel sel = tree_step_body<16, 1>(t, sim, mode, b, lane);
    if (!(mode & kTreeSelect)) return;
    const int node = __builtin_amdgcn_readfirstlane(sel.node), action = __builtin_amdgcn_readfirstlane(sel.action);
    const int dst = __builtin_amdgcn_readfirstlane(sel.dst);
    char* base = t.embeddings + static_cast<long>(b) * t.N * t.state_bytes;
    // embedding slots 1.. are only ever written here and the allocation starts zeroed: bytes >= 16 of `memory` stay zero
    subleq16_wave_body(e, base + static_cast<long>(node) * t.state_bytes, base + static_cast<long>(dst) * t.state_bytes, b, lane, action, 0,
                       Key{0, 0}, nullptr, nullptr, next_reward, next_term, next_legal, 16, obs_out, nf);
}

// The same for the word sizes whose tree is walked by a whole wavefront (64 <= word_size <= 256: 64 lanes x APL actions): tree_step_kernel<64, APL>
// followed by subleq_wave_kernel (program in the lanes of one VGPR) in one launch, the binary observation row and -- where the hashed slice is
// the IO rows -- its novelty included.  word_size 256 is the reference's code default (main.py:159): a simulation is this launch and the forward.
// The arguments are ONE struct read through the laundered kernarg segment pointer, once per phase (the trick of the one-launch kernels,
// search_subleq.hpp): passed by value, TreeDev + EnvDev + the output pointers were all loaded at the kernel entry and held in scalar
// registers across both phases -- 142 SGPR spills into VGPR lanes in the first version, which then ran no faster than the two launches.
struct SqTreeLeafArgs {
    TreeDev t;
    int sim, mode;
    EnvDev e;
    float* next_reward;
    uint8_t *next_term, *next_legal, *obs_out;
    NoveltyFuse nf;
};
__device__ __forceinline__ const SqTreeLeafArgs* sq_tree_leaf_args() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const SqTreeLeafArgs*)p;
}
template <int APL>
__global__ void __launch_bounds__(64) subleq_tree_leaf_kernel(SqTreeLeafArgs) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int node, action, dst;
    {
        const SqTreeLeafArgs& a = *sq_tree_leaf_args();
        const TreeSel sel = tree_step_body<64, APL>(a.t, a.sim, a.mode, b, lane);
        if (!(a.mode & kTreeSelect)) return;
        node = __builtin_amdgcn_readfirstlane(sel.node); action = __builtin_amdgcn_readfirstlane(sel.action);
        dst = __builtin_amdgcn_readfirstlane(sel.dst);
    }
    {
        const SqTreeLeafArgs& a = *sq_tree_leaf_args();
        char* base = a.t.embeddings + static_cast<long>(b) * a.t.N * a.t.state_bytes;
        subleq_wave_body(a.e, base + static_cast<long>(node) * a.t.state_bytes, base + static_cast<long>(dst) * a.t.state_bytes, b, lane, action, 0, Key{0, 0},
                         nullptr, nullptr, a.next_reward, a.next_term, a.next_legal, a.obs_out, a.nf);
    }
}

}  // namespace eaz
