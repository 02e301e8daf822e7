// tree.hpp -- Gumbel-MuZero search tree with epistemic fields: device layout and the per-simulation tree kernel.
//
// Reference: emctx [EXT] (fork of google-deepmind/mctx, un-vendored) as called from
// src/selfplay.py:100-121, with the recurrent fn of src/context.py:117-155 applied in the expand step.
//
// Execution model: a group of LPT (2..64, power of two) consecutive lanes of one wavefront walks ONE tree; each lane
// owns APL consecutive actions (LPT*APL >= A).  Small action sets therefore pack 64/LPT trees per wavefront
// (DeepSea: up to 32 trees per wave; the launch spreads them thinner, see tree_step_kernel); A = 256 uses a full wave with 4
// actions per lane.  All per-node vectors
// ([A]-sized rows of the [B,N,A] arrays) are read coalesced by the group; float reductions use the fixed
// butterfly order (lane ^ 1, ^ 2, ...) that the CPU oracle mirrors, so selections are bit-reproducible.
#pragma once
#include "dmath.hpp"
#include "envs.hpp"

namespace eaz {

struct TreeDev {
    int B, N, A, S;
    int32_t *node_visits, *parents, *action_from_parent;                                     // [B,N]
    float *raw_values, *node_values, *raw_var, *node_var;                                    // [B,N]
    int32_t *children_index, *children_visits;                                               // [B,N,A]
    float *children_prior_logits, *children_values, *children_rewards, *children_discounts;  // [B,N,A]
    float *children_var, *children_reward_var;                                               // [B,N,A]
    float* beta;            // [B]
    uint8_t* root_invalid;  // [B,A]
    float* root_gumbel;     // [B,A]
    char* embeddings;       // [B,N,state_bytes]
    int state_bytes;
    int32_t *sim_parent, *sim_action, *sim_next;  // [B] result of the last select
    int32_t *path_node, *path_action;             // [B,S] root-to-leaf path of the last select (level d: node, action taken)
    int32_t* path_len;                            // [B]   number of levels of that path (>= 1)
    const int32_t* table;                         // [(max_considered+1), S]
    // outputs of a caller-evaluated recurrent fn for the simulation being expanded (generic path, fused_rf == 0)
    const float *rf_reward, *rf_reward_var, *rf_discount, *rf_logits, *rf_value, *rf_value_var;  // [B], [B,A]
    // fused native recurrent fn (src/context.py:117-155): raw network outputs + fields of the stepped state
    int fused_rf;
    const float *net_logits, *net_value, *net_ube;  // [B,A] (head chosen by `exploration`), [B], [B] (u * max_u, pre-novelty)
    const float* net_novelty;                        // [B]   scaled_state_novelty of the stepped state (hash kernel)
    float max_u;                                     //       fully_connected.py:19
    const float* next_reward;                        // [B]   rewards[b, current_player]   context.py:139
    const uint8_t* next_term;                        // [B]   state.terminated             context.py:140-144
    const uint8_t* next_legal;                       // [B,A] state.legal_action_mask      context.py:137
    float discount;
    int two_player;
    // fused leaf step (fuse_env == 1 + EAZ_ENV_DEEPSEA): the select phase also applies env.step to the chosen (parent, action),
    // stores the child state in its embedding slot and hands the network the flat index of the one-hot observation -- no
    // leaf-gather launch, no materialised observation row (context.py:125-127 for an environment whose step is a few integer ops)
    int fuse_env;
    EnvDev fenv;
    float* fuse_reward;     // = next_reward
    uint8_t* fuse_term;     // = next_term
    int32_t* leaf_hot;      // [B] ds_hot_index of the stepped state
    // options
    int rescale, use_mixed, max_considered, max_depth;
    float value_scale, maxvisit_init;
    int bonus_after, var_backup;  // eaz_search_opts::epistemic_bonus_after_rescale / epistemic_variance_backup
    // lazy child rows (the per-simulation self-play driver on wide trees, eaz_api.hip selfplay_lazy_rows): of the eight [B,N,A] arrays only
    // the children_visits row of a node is initialised -- by the kernel that creates the node (tree_root_kernel / the expand) -- and every
    // reader takes mctx's initial value (index -1, floats 0) for an edge whose visit count is 0.  children_index[p,a] != -1 <=> visits > 0
    // between two simulations (the expand links the edge, the backward of the same launch counts it), so the results are those of cleared
    // arrays, bit for bit, without the 532 KB per tree (A = 256, N = 65) that tree_clear_kernel writes per search.
    int lazy_rows;
};

// running statistic of the variance backup (A.6 twin; include/eaz.h epistemic_variance_backup)
__device__ __forceinline__ float var_backup_update(int mode, float nvar, float cnt, float U) {
    if (mode == 0) return (nvar * cnt + U) / (cnt + 1.0f);
    return ((nvar * cnt) * cnt + U) / ((cnt + 1.0f) * (cnt + 1.0f));
}

// ---- cross-lane exchange ------------------------------------------------------------------------------------
// lane ^ MASK exchange.  MASK 1 and 2 are DPP quad permutes (no LDS round trip, ~8 cycles); 4 and 8 are two DPP moves each --
// gfx9 has no xor-mask DPP, but i ^ 4 = quad-reverse(half-mirror(i)) (i ^ 7 ^ 3) and i ^ 8 = half-mirror(row-mirror(i)) (i ^ 15 ^ 7)
// inside a row of 16 lanes -- instead of a ds_swizzle through the LDS crossbar (~4x the latency); 16 is ds_swizzle bit-mask mode,
// 32 falls back to ds_bpermute.  The reductions below run up to log2(LPT) of these per call and a dozen calls per tree level, so their
// latency is on the critical path of every simulation.
template <int MASK>
__device__ __forceinline__ int xor_lane_i(int v) {
    if constexpr (MASK == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4)
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);   // row_half_mirror, quad_perm [3,2,1,0]
    else if constexpr (MASK == 8)
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true), 0x141, 0xF, 0xF, true);  // row_mirror, row_half_mirror
    else if constexpr (MASK < 32) return __builtin_amdgcn_ds_swizzle(v, (MASK << 10) | 0x1F);  // and 0x1F, or 0, xor MASK
    else return __shfl_xor(v, MASK);
}
template <int MASK>
__device__ __forceinline__ float xor_lane_f(float v) {
    return __builtin_bit_cast(float, xor_lane_i<MASK>(__builtin_bit_cast(int, v)));
}

// ---- group reductions ---------------------------------------------------------------------------------------
#define EAZ_FOR_MASKS(LPT, BODY)                      \
    do {                                              \
        if constexpr (LPT > 1) { constexpr int M = 1; BODY }   \
        if constexpr (LPT > 2) { constexpr int M = 2; BODY }   \
        if constexpr (LPT > 4) { constexpr int M = 4; BODY }   \
        if constexpr (LPT > 8) { constexpr int M = 8; BODY }   \
        if constexpr (LPT > 16) { constexpr int M = 16; BODY } \
        if constexpr (LPT > 32) { constexpr int M = 32; BODY } \
    } while (0)
template <int LPT>
__device__ __forceinline__ float gsum(float v) {
    EAZ_FOR_MASKS(LPT, v = v + xor_lane_f<M>(v););
    return v;
}
template <int LPT>
__device__ __forceinline__ int gsumi(int v) {
    EAZ_FOR_MASKS(LPT, v = v + xor_lane_i<M>(v););
    return v;
}
template <int LPT>
__device__ __forceinline__ float gmax(float v) {
    EAZ_FOR_MASKS(LPT, { const float o = xor_lane_f<M>(v); v = o > v ? o : v; });
    return v;
}
template <int LPT>
__device__ __forceinline__ float gmin(float v) {
    EAZ_FOR_MASKS(LPT, { const float o = xor_lane_f<M>(v); v = o < v ? o : v; });
    return v;
}
template <int LPT>
__device__ __forceinline__ int gmaxi(int v) {
    EAZ_FOR_MASKS(LPT, { const int o = xor_lane_i<M>(v); v = o > v ? o : v; });
    return v;
}
// argmax with ties -> lowest index (jnp.argmax)
template <int LPT>
__device__ __forceinline__ int gargmax(float v, int i) {
    EAZ_FOR_MASKS(LPT, {
        const float ov = xor_lane_f<M>(v);
        const int oi = xor_lane_i<M>(i);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    });
    return i;
}
// in-lane pairwise sum of APL slots (levels below the cross-lane butterfly)
template <int APL>
__device__ __forceinline__ float lsum(const float (&x)[APL]) {
    if constexpr (APL == 1) return x[0];
    else if constexpr (APL == 2) return x[0] + x[1];
    else return (x[0] + x[1]) + (x[2] + x[3]);
}

template <int LPT, int APL>
struct Group {
    int b;      // tree index
    int gl;     // lane within the group
    int A;
    bool ok[APL];  // slot holds a real action
    int act[APL];

    __device__ __forceinline__ float sum(const float (&x)[APL]) const { return gsum<LPT>(lsum<APL>(x)); }
    __device__ __forceinline__ float vmax(const float (&x)[APL]) const {
        float m = -kInf;
#pragma unroll
        for (int j = 0; j < APL; ++j) if (ok[j]) m = x[j] > m ? x[j] : m;
        return gmax<LPT>(m);
    }
    __device__ __forceinline__ float vmin(const float (&x)[APL]) const {
        float m = kInf;
#pragma unroll
        for (int j = 0; j < APL; ++j) if (ok[j]) m = x[j] < m ? x[j] : m;
        return gmin<LPT>(m);
    }
    __device__ __forceinline__ int argmax(const float (&x)[APL]) const {
        float bv = -kInf;
        int bi = 0x7FFFFFFF;
#pragma unroll
        for (int j = 0; j < APL; ++j)
            if (ok[j] && (bi == 0x7FFFFFFF || x[j] > bv)) { bv = x[j]; bi = act[j]; }
        if (bi == 0x7FFFFFFF) bv = -kInf;
        // an all -inf row must return 0: a lane without real actions keeps index INT_MAX and value -inf
        return gargmax<LPT>(bv, bi);
    }
    // jax.nn.softmax over the real actions
    __device__ __forceinline__ void softmax(const float (&x)[APL], float (&p)[APL]) const {
        const float m = vmax(x);
        float e[APL];
        if constexpr (LPT == 1 && APL == 2) {
            // two real actions held by one lane: the larger one's exponent is x - m = +0 and det_exp(+0) is exactly 1.0f (n = 0, r = 0,
            // polynomial 0 + 1, scale 2^0), so one det_exp serves both -- same bits as the general form below
            if (ok[0] && ok[1]) {
                const float lo = x[0] < x[1] ? x[0] : x[1];
                const float el = det_exp(lo - m);
                e[0] = x[0] < x[1] ? el : 1.0f;
                e[1] = x[1] < x[0] ? el : 1.0f;
                const float s2 = e[0] + e[1];
                p[0] = e[0] / s2;
                p[1] = e[1] / s2;
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < APL; ++j) e[j] = ok[j] ? det_exp(x[j] - m) : 0.0f;
        const float s = sum(e);
#pragma unroll
        for (int j = 0; j < APL; ++j) p[j] = e[j] / s;
    }
};

// mctx qtransforms._compute_mixed_value [EXT] (A.5)
template <int LPT, int APL>
__device__ __forceinline__ float mixed_value(const Group<LPT, APL>& g, float raw, const float (&q)[APL],
                                             const int (&visits)[APL], const float (&probs)[APL], int sumN) {
    float t0[APL], t1[APL], p2[APL];
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        p2[j] = probs[j] > kF32Tiny ? probs[j] : kF32Tiny;
        t0[j] = (g.ok[j] && visits[j] > 0) ? p2[j] : 0.0f;
    }
    const float sum_probs = g.sum(t0);
#pragma unroll
    for (int j = 0; j < APL; ++j) t1[j] = (g.ok[j] && visits[j] > 0) ? (p2[j] * q[j]) / sum_probs : 0.0f;
    const float wq = g.sum(t1);
    return (raw + static_cast<float>(sumN) * wq) / (static_cast<float>(sumN) + 1.0f);
}

// per-node child rows held by a group
template <int APL>
struct NodeRows {
    float logits[APL], values[APL], rewards[APL], discounts[APL], var[APL], rvar[APL];
    int visits[APL], index[APL];
};

template <int LPT, int APL>
__device__ __forceinline__ void load_rows(const TreeDev& t, const Group<LPT, APL>& g, int node, bool need_var,
                                          NodeRows<APL>& r) {
    const long base = (static_cast<long>(g.b) * t.N + node) * t.A;
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        if (g.ok[j]) {
            const long o = base + g.act[j];
            r.logits[j] = t.children_prior_logits[o];
            r.values[j] = t.children_values[o];
            r.rewards[j] = t.children_rewards[o];
            r.discounts[j] = t.children_discounts[o];
            r.visits[j] = t.children_visits[o];
            r.index[j] = t.children_index[o];
            r.var[j] = need_var ? t.children_var[o] : 0.0f;
            r.rvar[j] = need_var ? t.children_reward_var[o] : 0.0f;
            if (t.lazy_rows && r.visits[j] == 0) {  // never linked since the node was created: the other rows hold whatever the allocation held
                r.values[j] = 0.0f; r.rewards[j] = 0.0f; r.discounts[j] = 0.0f;
                r.index[j] = -1; r.var[j] = 0.0f; r.rvar[j] = 0.0f;
            }
        } else {
            r.logits[j] = 0.0f; r.values[j] = 0.0f; r.rewards[j] = 0.0f; r.discounts[j] = 0.0f;
            r.visits[j] = 0; r.index[j] = -1; r.var[j] = 0.0f; r.rvar[j] = 0.0f;
        }
    }
}

// emctx epistemic_qtransform_completed_by_mix_value [EXT] (selfplay.py:114-116; A.5).  beta == 0 -> plain mctx.
// `probs` = jax.nn.softmax(prior_logits) of the node: a function of the node's logits alone, so a caller may compute it once per node
// (search_subleq_persist.hpp keeps it beside the logits) instead of once per visit -- same operations, same bits.
template <int LPT, int APL>
__device__ __forceinline__ void qtransform_p(const TreeDev& t, const Group<LPT, APL>& g, const NodeRows<APL>& r, const float (&probs)[APL],
                                             float raw_value, float raw_var, float beta, float (&out)[APL]) {
    float q[APL];
#pragma unroll
    for (int j = 0; j < APL; ++j) q[j] = r.rewards[j] + r.discounts[j] * r.values[j];
    int ln = 0, lmax = 0;
#pragma unroll
    for (int j = 0; j < APL; ++j) { ln += r.visits[j]; lmax = r.visits[j] > lmax ? r.visits[j] : lmax; }
    const int sumN = gsumi<LPT>(ln);
    const int maxN = gmaxi<LPT>(lmax);
    float value = raw_value;
    if (t.use_mixed) value = mixed_value<LPT, APL>(g, raw_value, q, r.visits, probs, sumN);
#pragma unroll
    for (int j = 0; j < APL; ++j) out[j] = r.visits[j] > 0 ? q[j] : value;
    float bonus[APL];
#pragma unroll
    for (int j = 0; j < APL; ++j) bonus[j] = 0.0f;
    if (beta != 0.0f) {  // x + 0*sqrt(v) == x for finite v >= 0, so the twin is only evaluated when it matters
        float qv[APL];
#pragma unroll
        for (int j = 0; j < APL; ++j) qv[j] = r.rvar[j] + (r.discounts[j] * r.discounts[j]) * r.var[j];
        float vvar = raw_var;
        if (t.use_mixed) vvar = mixed_value<LPT, APL>(g, raw_var, qv, r.visits, probs, sumN);
#pragma unroll
        for (int j = 0; j < APL; ++j) {
            const float cvar = r.visits[j] > 0 ? qv[j] : vvar;
            bonus[j] = beta * __builtin_sqrtf(cvar);
            if (!t.bonus_after) out[j] = out[j] + bonus[j];
        }
    }
    if (t.rescale) {
        const float mn = g.vmin(out), mx = g.vmax(out);
        float den = mx - mn;
        den = den > 1e-8f ? den : 1e-8f;
#pragma unroll
        for (int j = 0; j < APL; ++j) out[j] = (out[j] - mn) / den;
    }
    if (beta != 0.0f && t.bonus_after) {
#pragma unroll
        for (int j = 0; j < APL; ++j) out[j] = out[j] + bonus[j];
    }
    const float scale = (t.maxvisit_init + static_cast<float>(maxN)) * t.value_scale;
#pragma unroll
    for (int j = 0; j < APL; ++j) out[j] = scale * out[j];
}
template <int LPT, int APL>
__device__ __forceinline__ void qtransform(const TreeDev& t, const Group<LPT, APL>& g, const NodeRows<APL>& r,
                                           float raw_value, float raw_var, float beta, float (&out)[APL]) {
    float probs[APL];
    g.softmax(r.logits, probs);
    qtransform_p<LPT, APL>(t, g, r, probs, raw_value, raw_var, beta, out);
}

// mctx seq_halving.score_considered + masked_argmax [EXT] (A.2)
// gumbel / invalid: this lane's slots of root_gumbel[b, :] / root_invalid[b, :] (loaded by the caller ahead of the walk)
template <int LPT, int APL>
__device__ __forceinline__ int root_pick(const TreeDev& t, const Group<LPT, APL>& g, const NodeRows<APL>& r,
                                         const float (&cq)[APL], float considered_visit, const float (&gumbel)[APL],
                                         const uint8_t (&invalid)[APL]) {
    const float m = g.vmax(r.logits);
    float sc[APL];
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        if (g.ok[j]) {
            float s = (gumbel[j] + (r.logits[j] - m)) + cq[j];
            s = s > -1e9f ? s : -1e9f;
            s = s + (static_cast<float>(r.visits[j]) == considered_visit ? 0.0f : -kInf);
            sc[j] = invalid[j] ? -kInf : s;
        } else {
            sc[j] = -kInf;
        }
    }
    return g.argmax(sc);
}

template <int LPT, int APL>
__device__ __forceinline__ Group<LPT, APL> make_group(int b, int gl, int A) {
    Group<LPT, APL> g;
    g.b = b; g.gl = gl; g.A = A;
#pragma unroll
    for (int j = 0; j < APL; ++j) { g.act[j] = gl * APL + j; g.ok[j] = g.act[j] < A; }
    return g;
}

enum : int { kTreeExpand = 1, kTreeSelect = 2 };

#ifdef EAZ_FC_STAMPS
// Diagnostic build only (make diag): shader-clock stamps of thread 0 of every workgroup at the phase boundaries.
__device__ unsigned long long g_tree_stamps[1024 * 8];
#define EAZ_TSTAMP(i, extra)                                                                            \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 1024 && (mode & kTreeSelect)) {                           \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            g_tree_stamps[blockIdx.x * 8 + (i)] = t_;                                                  \
            if ((i) == 4) g_tree_stamps[blockIdx.x * 8 + 5] = (extra);                                 \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#else
#define EAZ_TSTAMP(i, extra)
#endif

// One launch per simulation: [expand + backward of simulation sim-1] then [select of simulation sim].
// `spread`: every tree owns LPT * spread consecutive lanes of which only the first LPT store (the others mirror them).  The
// kernel is bound by memory-level parallelism, not by lanes: one wavefront of 2-lane groups walks 32 trees whose rows lie in
// ~10 different cache lines per level each, so spreading the trees over more wavefronts / CUs (more miss queues) is what
// shortens a launch (DESIGN.md "tree_step_kernel").
// result of the select phase (group-uniform): the leaf's parent, the action taken there, the node slot of the child
struct TreeSel {
    int node, action, dst;
};

// The body of tree_step_kernel for tree `b`, lane `sub` of its LPT * spread lanes (also called by the fused tree + leaf kernel of
// search_kernels.hpp, one wavefront per tree).
// `mirror`: the tree a group beyond the batch walks (without storing): tree 0 in the per-simulation launches, where every group of the grid is
// in the same simulation; the one-launch search passes a tree of its OWN workgroup (another workgroup's tree would be read at whatever
// simulation that workgroup happens to be in).
template <int LPT, int APL>
__device__ __forceinline__ TreeSel tree_step_body(const TreeDev& t, int sim, int mode, int b, int sub, int mirror = 0) {
    TreeSel sel{0, 0, 0};
    const bool live = b < t.B && sub < LPT;  // mirror / dead groups keep executing (shuffles need every lane) without storing
    const Group<LPT, APL> g = make_group<LPT, APL>(b < t.B ? b : mirror, sub % LPT, t.A);
    const int N = t.N, A = t.A;
    const long nb = static_cast<long>(g.b) * N;
    const float beta = t.beta[g.b];
    const bool lead = live && g.gl == 0;
    EAZ_TSTAMP(0, 0);

    if (mode & kTreeExpand) {
        // ---- expand: mctx update_tree_node + topology (A.3) -------------------------------------------------
        // Every load that depends on the tree id only is issued before anything is consumed (one memory round trip):
        // the select result, the recorded path's first 8 levels (root first), the recurrent-fn outputs.
        constexpr int kChunk = 8;
        const long pb = static_cast<long>(g.b) * t.S;
        const int parent = t.sim_parent[g.b], action = t.sim_action[g.b], nxt = t.sim_next[g.b];
        // A group beyond the batch mirrors another tree without storing.  Round 2's one-launch Subleq search mirrored tree 0, which belongs to
        // ANOTHER workgroup that may not have finished its first select yet: the mirror then read the array's initial content, and a leftover
        // 0x7F...... of a freed allocation turned the loop below into 2^28 chunks -- a 257 s launch with correct results (round 3: rocprofv3
        // trace of the GPU suite).  Now: mirrors stay inside their workgroup, the arrays are zero-filled at creation, and the
        // depth is clamped.
        int D = t.path_len[g.b];
        D = D < 0 ? 0 : (D > t.S ? t.S : D);
        int p0n[kChunk], p0a[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            const int d = j < t.S ? j : t.S - 1;
            p0n[j] = t.path_node[pb + d];
            p0a[j] = t.path_action[pb + d];
        }
        float value, var, reward, rvar, disc, lg[APL];
        if (t.fused_rf) {
            // epistemic_recurrent_fn, src/context.py:132-153, on the outputs of the leaf-gather and network kernels
            float lraw[APL];
            uint8_t legal[APL];
#pragma unroll
            for (int j = 0; j < APL; ++j) {
                const long o = static_cast<long>(g.b) * A + (g.ok[j] ? g.act[j] : 0);
                lraw[j] = t.net_logits[o];
                legal[j] = t.next_legal[o];
            }
            const bool term = t.next_term[g.b] != 0;
            const float rw = t.next_reward[g.b], nval = t.net_value[g.b], nube = t.net_ube[g.b], nnov = t.net_novelty[g.b];
#pragma unroll
            for (int j = 0; j < APL; ++j) lg[j] = g.ok[j] ? lraw[j] : 0.0f;
            const float m = g.vmax(lg);                                                        // :135
#pragma unroll
            for (int j = 0; j < APL; ++j)
                if (g.ok[j]) lg[j] = legal[j] ? lg[j] - m : kF32Min;                           // :137
            reward = rw;                                                                       // :139
            value = term ? 0.0f : nval;                                                        // :140
            var = term ? 0.0f : ube_combine(nube, nnov, t.max_u);                              // :141 (+ fully_connected.py:92-96)
            disc = 1.0f * t.discount;                                                          // :142
            if (t.two_player) disc = disc * -1.0f;                                             // :143
            if (term) disc = 0.0f;                                                             // :144
            rvar = 0.0f;                                                                       // :149
        } else {
            value = t.rf_value[g.b]; var = t.rf_value_var[g.b];
            reward = t.rf_reward[g.b]; rvar = t.rf_reward_var[g.b]; disc = t.rf_discount[g.b];
#pragma unroll
            for (int j = 0; j < APL; ++j) lg[j] = t.rf_logits[static_cast<long>(g.b) * A + (g.ok[j] ? g.act[j] : 0)];
        }
#pragma unroll
        for (int j = 0; j < APL; ++j)
            if (live && g.ok[j]) {
                t.children_prior_logits[(nb + nxt) * A + g.act[j]] = lg[j];
                // lazy rows: a NEW node (the expand of simulation sim - 1 creates node `sim`; a node re-expanded at max_depth keeps its counts)
                if (t.lazy_rows && nxt == sim) t.children_visits[(nb + nxt) * A + g.act[j]] = 0;
            }
        const long ca0 = (nb + parent) * A + action;
        if (lead) {
            t.raw_values[nb + nxt] = value;
            t.raw_var[nb + nxt] = var;
            t.node_visits[nb + nxt] += 1;
            t.children_index[ca0] = nxt;
            t.children_rewards[ca0] = reward;
            t.children_reward_var[ca0] = rvar;
            t.children_discounts[ca0] = disc;
            t.parents[nb + nxt] = parent;
            t.action_from_parent[nb + nxt] = action;
        }
        EAZ_TSTAMP(1, 0);
        // ---- backward (A.6) ------------------------------------------------------------------------------------
        // The select of this simulation recorded its root-to-leaf path, so nothing is pointer-chased here: the
        // statistics of 8 consecutive levels are fetched with independent, unconditional loads (clamped indices: one
        // memory round trip per chunk), then folded leaf-to-root in registers.  Every lane computes, the lead lane stores.
        // Paths of <= 8 levels (the common case) use the entries prefetched above.
        float G = value, U = var;                    // leaf_value / leaf variance
        float child_value = value, child_var = var;  // node_values[idx] after its own update
        if (lead) { t.node_values[nb + nxt] = value; t.node_var[nb + nxt] = var; }
        for (int c = (D - 1) / kChunk; c >= 0; --c) {
            int pn[kChunk], pa[kChunk], cnt_i[kChunk], cvis[kChunk];
            float nv[kChunk], nvar[kChunk], er[kChunk], erv[kChunk], ed[kChunk];
            if (c > 0) {
#pragma unroll
                for (int j = 0; j < kChunk; ++j) {
                    const int d = c * kChunk + j < t.S ? c * kChunk + j : t.S - 1;
                    pn[j] = t.path_node[pb + d];
                    pa[j] = t.path_action[pb + d];
                }
            } else {
#pragma unroll
                for (int j = 0; j < kChunk; ++j) { pn[j] = p0n[j]; pa[j] = p0a[j]; }
            }
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                // levels beyond the path hold stale or unwritten entries: clamp, load, ignore
                pn[j] = pn[j] < 0 ? 0 : (pn[j] >= N ? N - 1 : pn[j]);
                pa[j] = pa[j] < 0 ? 0 : (pa[j] >= A ? A - 1 : pa[j]);
                const long ca = (nb + pn[j]) * A + pa[j];
                cnt_i[j] = t.node_visits[nb + pn[j]];
                nv[j] = t.node_values[nb + pn[j]];
                nvar[j] = t.node_var[nb + pn[j]];
                cvis[j] = t.children_visits[ca];
                er[j] = t.children_rewards[ca];
                erv[j] = t.children_reward_var[ca];
                ed[j] = t.children_discounts[ca];
            }
            // all reads of this chunk complete before the lead lane's stores below (same addresses, other lanes)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
            for (int j = kChunk - 1; j >= 0; --j) {
                const int d = c * kChunk + j;
                if (d < D) {
                    // the edge expanded by this very launch was stored by the lead lane a moment ago: take it from registers
                    const float r = d == D - 1 ? reward : er[j], rv = d == D - 1 ? rvar : erv[j], dd = d == D - 1 ? disc : ed[j];
                    const long ca = (nb + pn[j]) * A + pa[j];
                    const float cnt = static_cast<float>(cnt_i[j]);
                    G = r + dd * G;
                    U = rv + (dd * dd) * U;
                    const float pv = (nv[j] * cnt + G) / (cnt + 1.0f);
                    const float pvar = var_backup_update(t.var_backup, nvar[j], cnt, U);
                    if (lead) {
                        t.node_values[nb + pn[j]] = pv;
                        t.node_var[nb + pn[j]] = pvar;
                        t.node_visits[nb + pn[j]] = cnt_i[j] + 1;
                        t.children_values[ca] = child_value;
                        t.children_var[ca] = child_var;
                        t.children_visits[ca] = cvis[j] + 1;
                    }
                    child_value = pv;
                    child_var = pvar;
                }
            }
        }
        // make the lead lane's stores visible to the other lanes of the group before select re-reads the rows
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    EAZ_TSTAMP(2, 0);

    if (mode & kTreeSelect) {
        // ---- simulate (A.4) ----------------------------------------------------------------------------------
        int node = 0, depth = 0, action = 0, nxt = 0;
        const bool need_var = beta != 0.0f;
        // What the root's selection reads besides its rows depends on the tree id only: requested here, with the first row loads, instead
        // of one dependent round trip each inside the root iteration.
        float r_gumbel[APL];
        uint8_t r_invalid[APL];
        int lvalid0 = 0;
#pragma unroll
        for (int j = 0; j < APL; ++j) {
            const long o = static_cast<long>(g.b) * A + (g.ok[j] ? g.act[j] : 0);
            r_gumbel[j] = t.root_gumbel[o];
            r_invalid[j] = t.root_invalid[o];
            if (g.ok[j]) lvalid0 += 1 - static_cast<int>(r_invalid[j]);
        }
        while (true) {
            NodeRows<APL> r;
            load_rows<LPT, APL>(t, g, node, need_var, r);
            float cq[APL];
            qtransform<LPT, APL>(t, g, r, t.raw_values[nb + node], need_var ? t.raw_var[nb + node] : 0.0f, beta, cq);
            if (depth == 0) {
                // gumbel_muzero_root_action_selection
                int lsumN = 0;
#pragma unroll
                for (int j = 0; j < APL; ++j)
                    if (g.ok[j]) lsumN += r.visits[j];
                const int nvalid = gsumi<LPT>(lvalid0);
                int simidx = gsumi<LPT>(lsumN);
                const int nc = nvalid < t.max_considered ? nvalid : t.max_considered;
                simidx = simidx > t.S - 1 ? t.S - 1 : simidx;
                const float cvis = static_cast<float>(t.table[static_cast<long>(nc) * t.S + simidx]);
                action = root_pick<LPT, APL>(t, g, r, cq, cvis, r_gumbel, r_invalid);
            } else {
                // gumbel_muzero_interior_action_selection
                float x[APL], pr[APL];
                int lsumN = 0;
#pragma unroll
                for (int j = 0; j < APL; ++j) { x[j] = r.logits[j] + cq[j]; lsumN += r.visits[j]; }
                const int sumN = gsumi<LPT>(lsumN);
                g.softmax(x, pr);
                const float den = static_cast<float>(1 + sumN);
#pragma unroll
                for (int j = 0; j < APL; ++j) x[j] = pr[j] - static_cast<float>(r.visits[j]) / den;
                action = g.argmax(x);
            }
            // child index of the chosen action: already in the row registers of the lane that owns the action
            int mine = -2147483647 - 1;
#pragma unroll
            for (int j = 0; j < APL; ++j)
                if (g.ok[j] && g.act[j] == action) mine = r.index[j];
            nxt = gmaxi<LPT>(mine);
            if (lead) {
                t.path_node[static_cast<long>(g.b) * t.S + depth] = node;
                t.path_action[static_cast<long>(g.b) * t.S + depth] = action;
            }
            depth += 1;
            // group-uniform exit: every lane of the group computed the same action / nxt
            if (nxt == -1 || depth >= t.max_depth) break;
            node = nxt;
        }
        EAZ_TSTAMP(3, 0);
        if (lead) {
            const int dst = nxt == -1 ? sim + 1 : nxt;
            t.sim_parent[g.b] = node;
            t.sim_action[g.b] = action;
            t.sim_next[g.b] = dst;
            t.path_len[g.b] = depth;
            if (t.fuse_env == 1 + EAZ_ENV_DEEPSEA) {
                char* base = t.embeddings + nb * static_cast<long>(sizeof(eaz_deepsea_state));
                const eaz_deepsea_state ns = ds_step(t.fenv, ds_load(base + static_cast<long>(node) * sizeof(eaz_deepsea_state)), action);
                ds_store(base + static_cast<long>(dst) * sizeof(eaz_deepsea_state), ns);
                t.fuse_reward[g.b] = ns.reward;       // rewards[b, current_player], single player (deep_sea.py:100-102)
                t.fuse_term[g.b] = ns.terminated;
                t.leaf_hot[g.b] = ds_hot_index(t.fenv, ns);
            }
        }
        EAZ_TSTAMP(4, static_cast<unsigned long long>(depth));
        sel.node = node;
        sel.action = action;
        sel.dst = nxt == -1 ? sim + 1 : nxt;
    }
    return sel;
}

template <int LPT, int APL>
__global__ void __launch_bounds__(256) tree_step_kernel(TreeDev t, int sim, int mode, int spread) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = LPT * spread;
    const int b = gtid / slot, sub = gtid - b * slot;
    tree_step_body<LPT, APL>(t, sim, mode, b, sub);
}

}  // namespace eaz
