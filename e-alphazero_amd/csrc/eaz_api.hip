// eaz_api.hip -- implementation of the C ABI declared in include/eaz.h (libeaz.so).
//
// Host-side orchestration only: object lifetimes, workspace allocation, kernel launches on the caller's stream.
// No torch types, no CPU fallback: every entry point either enqueues gfx950 kernels or fails with an error.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/eaz.h"
#include "env_kernels.hpp"
#include "net_fc.hpp"
#include "net_resnet.hpp"
#include "search_kernels.hpp"
#include "search_persist.hpp"
#include "net_fc_pipe.hpp"
#include "search_subleq.hpp"
#include "search_subleq_persist.hpp"

using namespace eaz;

// ------------------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define ARG_OK(cond)                                                       \
    do {                                                                   \
        if (!(cond)) return fail("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
    } while (0)
#define LAUNCH_OK() HIP_OK(hipGetLastError())

extern "C" const char* eaz_last_error(void) { return g_err.c_str(); }
extern "C" int eaz_version(void) { return 100; }
#ifndef EAZ_SRC_HASH
#define EAZ_SRC_HASH "unknown"
#endif
extern "C" const char* eaz_build_hash(void) { return EAZ_SRC_HASH; }

static inline int cdiv(long a, long b) { return static_cast<int>((a + b - 1) / b); }
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ------------------------------------------------------------------------------------------------------------------
// objects
// ------------------------------------------------------------------------------------------------------------------
struct eaz_ctx {
    int device;
};

struct eaz_env {
    eaz_ctx* ctx;
    EnvDev dev;
    uint8_t* d_action_map;
};

struct eaz_net {
    eaz_ctx* ctx;
    int kind;  // 0 = fully connected (fully_connected.py), 1 = residual (resnet.py)
    ResNetDev rdev;
    float* d_rparams;
    size_t rparams_count;
    NetDev dev;
    float* d_w[12];
    float* d_b[12];
    size_t bitset_bytes;
    uint32_t* d_ztab;                     // NetDev::ztab storage
    uint32_t* d_hot_index;                // NetDev::hot_index storage
    float* d_sim_matrix;                  // NetDev::sim_matrix storage (SimHash)
    unsigned long long* d_kstep_counter;  // executed layer-1 k-steps (profiling only; nullptr = off)
    unsigned long long* d_kstep_storage;
    int32_t* d_in_perm;     // NetDev::in_perm / in_inv storage ([2][in_dim]); null: ascending input order
    float* d_w_raw[4];      // layer-1 weights as uploaded (rows in the original order) while an input order is set
    std::vector<int32_t> perm;  // host copy of the order (empty: ascending)
    float* d_w1q[4];        // NetDev::w1q storage (built on the first Subleq-16 one-launch call, sq16_pack_w1)
    uint64_t w_stamp;       // counts weight uploads (create, eaz_net_set_weights)
    uint64_t w1q_stamp;     // the upload d_w1q was packed from
};

struct ProfSpan {
    hipEvent_t a, b;
    int cls;
};

struct eaz_search {
    eaz_ctx* ctx;
    eaz_env* env;
    eaz_net* net;
    eaz_search_opts opts;
    int B, N, A, S;
    int lpt, apl;
    TreeDev tree;
    std::vector<void*> allocs;
    // workspace of one simulation
    uint8_t* d_obs;
    uint8_t* d_obs_perm;  // [B, obs_dim]: the observation rows in the network's layer-1 input order (fc_launch scratch)
    float *d_novelty, *d_exploit, *d_explore, *d_v, *d_u, *d_next_reward;
    uint8_t *d_next_term, *d_next_legal;
    int32_t* d_leaf_hot;
    int fuse_leaf;  // DeepSea + FC network: env.step fused into the tree kernel, one-hot forward (TreeDev::fuse_env)
    int32_t* d_table;
    // root buffers used by the self-play driver
    float *d_root_logits, *d_root_beta, *d_root_gumbel, *d_aw, *d_vprobs;
    uint8_t *d_root_invalid, *d_legal;
    int32_t* d_action;
    // profiling
    int prof_on;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> ev_pool;
    // side stream for the novelty lookup (fork/join around every network forward)
    hipStream_t side, main;
    hipEvent_t ev_fork, ev_join, ev_in, ev_out;
    float graph_beta;
    // per-real-step staging (fixed addresses so that one captured hipGraph serves every step) + the graph itself
    float *d_root_v, *d_root_u, *d_sum_value, *d_sum_std, *d_sum_qvar, *d_gumbel_stage;
    float *d_sum_q, *d_sum_vc, *d_next_v, *d_reward_tmp;  // reanalyze / evaluate scratch
    uint8_t *d_term, *d_term2;
    int* d_alive;
    Key* d_step_keys;  // [4]: gumbel, env, sample-visits, sample-improved keys of the current real step
    struct StepOut* d_step_out;  // destination slots of the current real step's SelfplayOutput (read by the captured graph)
    hipGraphExec_t step_graph;
    const void* graph_states;  // the captured graph is valid for this d_states pointer / these flags
    int graph_flags;
    int pending_expand;  // simulation whose expand+backward has not run yet (-1: none)
    // persistent search (search_persist.hpp): one launch runs all S simulations of 16 trees per workgroup in LDS
    int persist_ok;      // DeepSea + FC network + 2 actions + the trees of a workgroup fit into LDS
    int persist_lds;     // dynamic LDS bytes of that launch
    int clear_pending;   // eaz_search_begin left the global tree arrays uncleared (the persistent kernel rewrites all of them)
    int sims_done;       // simulations run since eaz_search_begin
    unsigned long long* d_stamps;
    // per-simulation path, sub-batch pipelining: views of this search over [0, b0) and [b0, B) that run the simulations on two streams, so
    // that one half's latency-bound env-step kernel overlaps the other half's MFMA-bound forward (Subleq: the interpreter launch is one
    // wavefront per CU for 60 us).  Views share the parent's allocations (pointer offsets), own their streams / events.
    std::vector<eaz_search*> shards;
    hipEvent_t ev_shard_fork, ev_shard_join;
    PersistSelfplay last_sp;   // arguments of the last fused self-play launch (eaz_profile_replay)
    char* d_replay_states;     // scratch states for that replay
    // Subleq-16 fused self-play (search_subleq_persist.hpp): one launch per selfplay() call, trees and states in LDS
    int sqp_lds;               // dynamic LDS bytes of that launch (0: the trees of a workgroup do not fit)
    float* d_sq_logits;        // [ceil16(B), N, 32] scratch rows: prior logits | their softmax
    float* d_sq_l1init;        // [6 task branches, 3, 256] + [ceil16(B), 3, 256]: start vectors of the layer-1 chains (Sq16SelfplayArgs::l1init)
    int sqp_roots_fit;         // the roots of the search begun last fit the one-launch kernel's compact state (sq16_roots_fit)
    int ds_roots_fit;          // likewise for DeepSea (ds_roots_fit): the packed state word of search_persist.hpp
    int* h_fit_flag;           // pinned host word the two root checks write through its device mapping (no copy: launch + stream synchronisation)
    int* d_fit_flag;           // its device address
    int fit_epoch;             // value a misfit writes in the current check (no clear between checks)
    int* d_eval_wg;            // [1 + workgroups]: evaluate's trip count, then the workgroups' own (PersistSelfplay::eval_iters / eval_wg_iters)
};

extern "C" int eaz_ctx_create(int device, eaz_ctx** out) {
    ARG_OK(out != nullptr);
    int n = 0;
    HIP_OK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail("device %d out of range (%d visible)", device, n);
    HIP_OK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("libeaz is built for gfx950 (MI355X); device %d is %s", device, prop.gcnArchName);
    *out = new eaz_ctx{device};
    return 0;
}
extern "C" int eaz_ctx_destroy(eaz_ctx* ctx) {
    delete ctx;
    return 0;
}
extern "C" int eaz_key_fold(uint64_t key, uint32_t tag, uint64_t* out) {
    ARG_OK(out != nullptr);
    const Key k = fold(key_from_u64(key), tag);
    *out = static_cast<uint64_t>(k.k0) | (static_cast<uint64_t>(k.k1) << 32);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// environments
// ------------------------------------------------------------------------------------------------------------------
extern "C" int eaz_env_create_deepsea(eaz_ctx* ctx, int G, const uint8_t* h_action_map, eaz_env** out) {
    ARG_OK(ctx && out && G >= 2 && G <= 1024);
    HIP_OK(hipSetDevice(ctx->device));
    eaz_env* e = new eaz_env();
    e->ctx = ctx;
    std::memset(&e->dev, 0, sizeof e->dev);
    e->dev.kind = EAZ_ENV_DEEPSEA;
    e->dev.A = 2;
    e->dev.G = G;
    e->dev.obs_rows = G;
    e->dev.obs_cols = G;
    e->dev.obs_dim = G * G;
    e->dev.state_bytes = sizeof(eaz_deepsea_state);
    HIP_OK(hipMalloc(&e->d_action_map, static_cast<size_t>(G) * G));
    if (h_action_map) HIP_OK(hipMemcpy(e->d_action_map, h_action_map, static_cast<size_t>(G) * G, hipMemcpyHostToDevice));
    else HIP_OK(hipMemset(e->d_action_map, 0, static_cast<size_t>(G) * G));
    e->dev.action_map = e->d_action_map;
    *out = e;
    return 0;
}

extern "C" int eaz_env_create_subleq(eaz_ctx* ctx, int W, const int32_t* h_tasks, int n_tasks, int binary, eaz_env** out) {
    ARG_OK(ctx && out && h_tasks);
    if (W < 16 || W > 256) return fail("word_size must be in [16, 256] (subleq.py:606), got %d", W);
    ARG_OK(n_tasks >= 1 && n_tasks <= 16);
    eaz_env* e = new eaz_env();
    e->ctx = ctx;
    e->d_action_map = nullptr;
    std::memset(&e->dev, 0, sizeof e->dev);
    e->dev.kind = EAZ_ENV_SUBLEQ;
    e->dev.W = W;
    e->dev.A = W;
    e->dev.n_tasks = n_tasks;
    for (int i = 0; i < n_tasks; ++i) e->dev.tasks[i] = h_tasks[i];
    e->dev.binary = binary ? 1 : 0;
    int bl = 0;
    for (int v = W - 1; v > 0; v >>= 1) bl++;
    e->dev.obs_rows = W + 4 * EAZ_SUBLEQ_IO;          // subleq.py:618-621
    e->dev.obs_cols = binary ? bl + 1 : W + 1;        // subleq.py:59-60 / :613
    e->dev.obs_dim = e->dev.obs_rows * e->dev.obs_cols;
    e->dev.state_bytes = sizeof(eaz_subleq_state);
    *out = e;
    return 0;
}

extern "C" int eaz_env_set_subleq_reward(eaz_env* env, int reward_fn) {
    ARG_OK(env != nullptr);
    if (env->dev.kind != EAZ_ENV_SUBLEQ) return fail("eaz_env_set_subleq_reward: not a Subleq environment");
    if (reward_fn != EAZ_SUBLEQ_REWARD_SOLVED_OR_NOT && reward_fn != EAZ_SUBLEQ_REWARD_LOWEST_BYTES)
        return fail("eaz_env_set_subleq_reward: unknown reward function %d (subleq.py:535-542)", reward_fn);
    env->dev.reward_fn = reward_fn;
    return 0;
}

extern "C" int eaz_subleq16_test_vectors(int task, int c, uint16_t in[8], uint16_t expected_out[8]) {
    ARG_OK(in && expected_out && c >= 0 && c < 3);
    static constexpr Sq16CaseTable kHost = sq16_make_cases();  // the same constant expression the device table is initialised with
    const int branch = task - 1 < 0 ? 0 : (task - 1 > 5 ? 5 : task - 1);
    for (int i = 0; i < 8; ++i) {
        in[i] = static_cast<uint16_t>((kHost.v[branch][c][0] >> (8 * i)) & 0xFF);
        expected_out[i] = static_cast<uint16_t>((kHost.v[branch][c][1] >> (8 * i)) & 0xFF);
    }
    return 0;
}

extern "C" int eaz_subleq_test_vectors(int word_size, int task, int c, uint16_t in[8], uint16_t expected_out[8]) {
    ARG_OK(in && expected_out && c >= 0 && c < 3 && word_size >= 16 && word_size <= 256);
    subleq_test_case(task, c, word_size, in, expected_out);  // the function the per-lane interpreter kernel calls (csrc/envs.hpp)
    return 0;
}

extern "C" int eaz_env_create_connect4(eaz_ctx* ctx, eaz_env** out) {
    ARG_OK(ctx && out);
    eaz_env* e = new eaz_env();
    e->ctx = ctx;
    e->d_action_map = nullptr;
    std::memset(&e->dev, 0, sizeof e->dev);
    e->dev.kind = EAZ_ENV_CONNECT4;
    e->dev.A = 7;
    e->dev.obs_rows = 42;  // [6,7,2] flattened: 42 cells x 2 planes
    e->dev.obs_cols = 2;
    e->dev.obs_dim = 84;
    e->dev.state_bytes = sizeof(eaz_connect4_state);
    *out = e;
    return 0;
}

extern "C" int eaz_env_num_players(const eaz_env* env) { return env && env->dev.kind == EAZ_ENV_CONNECT4 ? 2 : 1; }

extern "C" int eaz_env_destroy(eaz_env* env) {
    if (!env) return 0;
    if (env->d_action_map) (void)hipFree(env->d_action_map);
    delete env;
    return 0;
}

extern "C" int eaz_env_info(const eaz_env* env, int* kind, int* num_actions, int* obs_rows, int* obs_cols, int* state_bytes) {
    ARG_OK(env != nullptr);
    if (kind) *kind = env->dev.kind;
    if (num_actions) *num_actions = env->dev.A;
    if (obs_rows) *obs_rows = env->dev.obs_rows;
    if (obs_cols) *obs_cols = env->dev.obs_cols;
    if (state_bytes) *state_bytes = env->dev.state_bytes;
    return 0;
}

// mode: 0 step, 1 auto-reset step, 2 init.  src/dst: plain [B] batches.
// mover_reward (optional): rewards[b, current_player before the step] of the new states
static int g_subleq_wave = [] { const char* e = std::getenv("EAZ_SUBLEQ_WAVE"); return e ? std::atoi(e) : 1; }();  // word sizes other than 16: wavefront per environment (1) or lane per environment (0)
// environments per one-wavefront workgroup of subleq_step_kernel: spread until the launch has a workgroup per CU
static int subleq_envs_per_wg(int B) {
    int epw = 64;
    while (epw > 8 && cdiv(B, epw) < 256) epw /= 2;
    return epw;
}
static int subleq_lds_bytes(int W) { return 2 * 64 * (((W + 3) & ~3) + 4); }  // program + snapshot per lane, rows of whole 32-bit words + one (bank spread)

static int env_step_launch(eaz_env* env, int B, const void* in, const int32_t* action, Key key, int mode, void* out,
                           void* out_copy, hipStream_t st, const Key* d_key = nullptr, float* mover_reward = nullptr) {
    if (B <= 0) return 0;
    StateRef src{const_cast<char*>(static_cast<const char*>(in ? in : out)), nullptr, 1};
    StateRef dst{static_cast<char*>(out), nullptr, 1};
    if (env->dev.kind == EAZ_ENV_DEEPSEA) {
        deepsea_step_kernel<<<cdiv(B, 256), 256, 0, st>>>(env->dev, static_cast<const char*>(in), action, B, mode,
                                                         static_cast<char*>(out), static_cast<char*>(out_copy));
        LAUNCH_OK();
        if (mover_reward) {
            env_fields_kernel<<<cdiv(B, 256), 256, 0, st>>>(env->dev, static_cast<const char*>(out), B, nullptr, nullptr, mover_reward, nullptr,
                                                           nullptr, nullptr);
            LAUNCH_OK();
        }
    } else if (env->dev.kind == EAZ_ENV_SUBLEQ) {
        if (env->dev.W == 16) {  // one wavefront per environment
            subleq16_wave_kernel<<<B, 64, 0, st>>>(env->dev, src, dst, action, B, mode, key, d_key, static_cast<char*>(out_copy), mover_reward,
                                                   nullptr, nullptr, 256, nullptr, NoveltyFuse{nullptr, nullptr, 0, 0.0f});
        } else {
            if (g_subleq_wave) {  // one wavefront per environment (default)
                subleq_wave_kernel<<<B, 64, 0, st>>>(env->dev, src, dst, action, B, mode, key, d_key, static_cast<char*>(out_copy), mover_reward, nullptr,
                                                     nullptr, nullptr, NoveltyFuse{nullptr, nullptr, 0, 0.0f});
            } else {
                const int epw = subleq_envs_per_wg(B);
                subleq_step_kernel<<<cdiv(B, epw), 64, subleq_lds_bytes(env->dev.W), st>>>(env->dev, src, dst, action, B, mode, key, d_key,
                                                                                          static_cast<char*>(out_copy), mover_reward, nullptr, nullptr,
                                                                                          epw, 256);
            }
        }
        LAUNCH_OK();
    } else {
        connect4_step_kernel<<<cdiv(B, 256), 256, 0, st>>>(env->dev, src, dst, action, B, mode, key, d_key, mover_reward, nullptr, nullptr);
        LAUNCH_OK();
        if (out_copy) HIP_OK(hipMemcpyAsync(out_copy, out, static_cast<size_t>(B) * env->dev.state_bytes, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

extern "C" int eaz_env_init(eaz_env* env, int B, uint64_t key, void* d_states, void* stream) {
    ARG_OK(env && B >= 0);
    if (B == 0) return 0;  // an empty batch is a no-op (a zero-element tensor has a null data pointer)
    ARG_OK(d_states != nullptr);
    return env_step_launch(env, B, nullptr, nullptr, key_from_u64(key), 2, d_states, nullptr, S(stream));
}

extern "C" int eaz_env_step(eaz_env* env, int B, const void* d_in, const int32_t* d_action, uint64_t key, int autoreset,
                            void* d_out, void* stream) {
    ARG_OK(env && B >= 0);
    if (B == 0) return 0;
    ARG_OK(d_in && d_action && d_out);
    return env_step_launch(env, B, d_in, d_action, key_from_u64(key), autoreset ? 1 : 0, d_out, nullptr, S(stream));
}

static int obs_launch(eaz_env* env, int B, const void* states, uint8_t* obs, hipStream_t st) {
    if (B <= 0) return 0;
    StateRef src{const_cast<char*>(static_cast<const char*>(states)), nullptr, 1};
    obs_emit_kernel<<<cdiv(static_cast<long>(B) * 64, 256), 256, 0, st>>>(env->dev, src, B, obs);
    LAUNCH_OK();
    return 0;
}

extern "C" int eaz_env_observe(eaz_env* env, int B, const void* d_states, uint8_t* d_obs, void* stream) {
    ARG_OK(env && B >= 0);
    if (B == 0) return 0;
    ARG_OK(d_states && d_obs);
    return obs_launch(env, B, d_states, d_obs, S(stream));
}

extern "C" int eaz_env_fields(eaz_env* env, int B, const void* d_states, uint8_t* d_term, uint8_t* d_trunc, float* d_reward,
                              int32_t* d_step_count, uint8_t* d_legal, int32_t* d_current_player, void* stream) {
    ARG_OK(env && B >= 0);
    if (B == 0) return 0;  // an empty batch is a no-op (a zero-element tensor has a null data pointer)
    ARG_OK(d_states != nullptr);
    if (B == 0) return 0;
    env_fields_kernel<<<cdiv(B, 256), 256, 0, S(stream)>>>(env->dev, static_cast<const char*>(d_states), B, d_term, d_trunc,
                                                         d_reward, d_step_count, d_legal, d_current_player);
    LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// network
// ------------------------------------------------------------------------------------------------------------------
static void layer_dims(int in_dim, int H, int A, int l, int* k, int* n) {
    *k = (l % 3) == 0 ? in_dim : H;
    *n = (l % 3) < 2 ? H : ((l / 3) < 2 ? 1 : A);
}

static int net_create_fc_impl(eaz_net* n, int in_dim, int hidden, int num_actions, const float* const* h_w, const float* const* h_b,
                              float max_ube, float max_reward_variance, int hash_off, int hash_len, int hash_bits) {
    std::memset(&n->dev, 0, sizeof n->dev);
    n->dev.in_dim = in_dim; n->dev.H = hidden; n->dev.A = num_actions;
    n->dev.max_u = max_ube; n->dev.nov_scale = max_reward_variance;
    n->dev.hash_off = hash_off; n->dev.hash_len = hash_len; n->dev.hash_bits = hash_bits;
    for (int l = 0; l < 12; ++l) {
        int k, m;
        layer_dims(in_dim, hidden, num_actions, l, &k, &m);
        // layer-1 weights are followed by zero rows up to a whole number of 16-k-step rounds: the pipelined forward (net_fc_pipe.hpp)
        // streams whole rounds and its reads stay inside the allocation
        const size_t rows = (l % 3) == 0 ? static_cast<size_t>(fp_kpad(k)) : static_cast<size_t>(k);
        HIP_OK(hipMalloc(&n->d_w[l], sizeof(float) * rows * m));
        if (rows != static_cast<size_t>(k)) HIP_OK(hipMemset(n->d_w[l], 0, sizeof(float) * rows * m));
        HIP_OK(hipMalloc(&n->d_b[l], sizeof(float) * m));
        HIP_OK(hipMemcpy(n->d_w[l], h_w[l], sizeof(float) * k * m, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(n->d_b[l], h_b[l], sizeof(float) * m, hipMemcpyHostToDevice));
        n->dev.w[l] = n->d_w[l];
        n->dev.b[l] = n->d_b[l];
    }
    n->bitset_bytes = static_cast<size_t>(1) << (hash_bits - 3);
    if (n->bitset_bytes < 4) n->bitset_bytes = 4;
    HIP_OK(hipMalloc(&n->dev.bitset, n->bitset_bytes));
    HIP_OK(hipMemset(n->dev.bitset, 0, n->bitset_bytes));
    {
        // hash lane i after n all-zero stripes (hashes.py:176-181,217-221 with lane value 0): n = 0 .. L
        const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, SEED = 1u;
        const int L = hash_len / 4;
        std::vector<uint32_t> zt(static_cast<size_t>(4) * (L + 1));
        const uint32_t init[4] = {SEED + P1 + P2, SEED + P2, SEED, SEED - P1};
        for (int i = 0; i < 4; ++i) {
            uint32_t acc = init[i];
            for (int k = 0; k <= L; ++k) {
                zt[static_cast<size_t>(i) * (L + 1) + k] = acc;
                acc = ((acc << 13) | (acc >> 19)) * P1;
            }
        }
        HIP_OK(hipMalloc(&n->d_ztab, zt.size() * sizeof(uint32_t)));
        HIP_OK(hipMemcpy(n->d_ztab, zt.data(), zt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        n->dev.ztab = n->d_ztab;
        // get_indices() (hashes.py:199-229) of every one-hot row: lane q holds the stripe with the 1.0f (bitcast, :207) at position n1,
        // the other lanes see zero stripes only; merge (:184-187), + stripe count (:226), avalanche (:190-197), top hash_bits (:229)
        const uint32_t P3 = 0xC2B2AE3Du;
        auto rotl = [](uint32_t x, int r) { return (x << r) | (x >> (32 - r)); };
        std::vector<uint32_t> hi(static_cast<size_t>(in_dim));
        for (int h = 0; h < in_dim; ++h) {
            uint32_t acc[4];
            for (int i = 0; i < 4; ++i) acc[i] = zt[static_cast<size_t>(i) * (L + 1) + L];
            const int hh = h - hash_off;
            if (hh >= 0 && hh < hash_len) {
                const int q = hh / L, n1 = hh - q * L;
                uint32_t x = rotl(zt[static_cast<size_t>(q) * (L + 1) + n1] + 0x3F800000u * P2, 13) * P1;
                for (int k = n1 + 1; k < L; ++k) x = rotl(x, 13) * P1;
                acc[q] = x;
            }
            uint32_t v = rotl(acc[0], 1) + rotl(acc[1], 7) + rotl(acc[2], 12) + rotl(acc[3], 18);
            v += static_cast<uint32_t>(L);
            v ^= v >> 15; v *= P2; v ^= v >> 13; v *= P3; v ^= v >> 16;
            hi[h] = hash_bits >= 32 ? v : v >> (32 - hash_bits);
        }
        HIP_OK(hipMalloc(&n->d_hot_index, hi.size() * sizeof(uint32_t)));
        HIP_OK(hipMemcpy(n->d_hot_index, hi.data(), hi.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        n->dev.hot_index = n->d_hot_index;
    }
    if (hidden == kFcH && fp_eligible(in_dim, num_actions))
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(fc16p_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, fp_lds_bytes(kFpMaxK, kFcW3Cols)));
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_forward_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kFcLdsBytes));
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(fc_forward_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kFcLdsBytes));
    HIP_OK(hipMalloc(&n->d_kstep_storage, 8 * sizeof(unsigned long long)));
    HIP_OK(hipMemset(n->d_kstep_storage, 0, 8 * sizeof(unsigned long long)));
    n->d_kstep_counter = nullptr;
    n->dev.l1q_mode = -1;
    for (int h = 0; h < 4; ++h) n->dev.w1raw[h] = n->d_w[h * 3];
    n->w_stamp = 1;
    return 0;
}

extern "C" int eaz_net_destroy(eaz_net* n);

extern "C" int eaz_net_create_fc(eaz_ctx* ctx, int in_dim, int hidden, int num_actions, const float* const* h_w,
                                 const float* const* h_b, float max_ube, float max_reward_variance, int hash_off,
                                 int hash_len, int hash_bits, eaz_net** out) {
    ARG_OK(ctx && out && h_w && h_b && in_dim > 0 && num_actions > 0 && num_actions <= 256);
    // linear_layer_size 256 (config.py:24, the only value the reference's configurations use) runs the MFMA kernels; any other size the plain
    // kernel of net_fc.hpp (fc_generic_forward_kernel): same results contract, per-simulation path only, not tuned
    if (hidden < 1 || hidden > 1024) return fail("linear_layer_size %d outside 1..1024", hidden);
    if (hash_len <= 0 || hash_len % 4 != 0 || hash_off < 0 || hash_off + hash_len > in_dim)
        return fail("hash slice [%d, %d) invalid: length must be a positive multiple of 4 (hashes.py:210) inside the observation", hash_off, hash_off + hash_len);
    ARG_OK(hash_bits >= 4 && hash_bits <= 32);
    if (in_dim > 256 * kFcMaxBlocks) return fail("observation size %d exceeds the FC kernel's limit %d", in_dim, 256 * kFcMaxBlocks);
    HIP_OK(hipSetDevice(ctx->device));
    eaz_net* n = new eaz_net();  // value-initialised: every pointer null, so a failed creation can be handed to eaz_net_destroy
    n->ctx = ctx;
    n->kind = 0;
    const int rc = net_create_fc_impl(n, in_dim, hidden, num_actions, h_w, h_b, max_ube, max_reward_variance, hash_off, hash_len, hash_bits);
    if (rc) { (void)eaz_net_destroy(n); return rc; }
    *out = n;
    return 0;
}

// haiku-ordered ResNet arrays (include/eaz.h) -> one packed float buffer + the offsets the kernel reads (ResNetDev).  Pure host work:
// used by eaz_net_create_resnet and, with identical offsets, by eaz_net_set_resnet_weights (in-place refresh after a learner update).
static void resnet_pack(const float* const* a, int H, int W, int Cin, int C, int nb, int A, ResNetDev& r, std::vector<float>& pk) {
    auto push = [&](const float* src, size_t cnt) { const int off = static_cast<int>(pk.size()); pk.insert(pk.end(), src, src + cnt); while (pk.size() % 4) pk.push_back(0.0f); return off; };
    std::memset(&r, 0, sizeof r);
    r.H = H; r.W = W; r.Cin = Cin; r.nb = nb; r.A = A; r.P = H * W;
    r.cin_log2 = 0;
    while ((1 << r.cin_log2) < Cin) r.cin_log2++;
    int i = 0, ci = 0, bi = 0;
    auto take_conv = [&](int taps, int cin, int cout, int& w_off, int& b_off, int* kpad, bool tower = false) {
        const int K = taps * cin, Kp = (K + 3) & ~3;
        w_off = push(a[i], static_cast<size_t>(K) * cout);
        pk.resize(static_cast<size_t>(w_off) + static_cast<size_t>(Kp) * cout, 0.0f);  // zero rows up to the padded K
        if (tower) {
            // the tower's 64 -> 64 convolutions in the kernel's operand order (net_resnet.hpp): four consecutive k-steps of a lane side by side
            const float* W0 = a[i];
            float* Wp = pk.data() + w_off;
            for (int ks4 = 0; ks4 < K / 16; ++ks4)
                for (int kq = 0; kq < 4; ++kq)
                    for (int c = 0; c < cout; ++c)
                        for (int q = 0; q < 4; ++q) Wp[((ks4 * 4 + kq) * cout + c) * 4 + q] = W0[static_cast<size_t>(4 * (4 * ks4 + q) + kq) * cout + c];
        }
        while (pk.size() % 4) pk.push_back(0.0f);
        b_off = push(a[i + 1], cout);
        if (kpad) *kpad = Kp;
        i += 2;
    };
    auto take_bn = [&](int c, int& mean_off, int& inv_off, int& off_off) {
        std::vector<float> inv(c);
        for (int k = 0; k < c; ++k) inv[k] = a[i][k] / std::sqrt(a[i + 3][k] + 1e-5f);  // scale / sqrt(var + eps), f32 like the oracle
        mean_off = push(a[i + 2], c); inv_off = push(inv.data(), c); off_off = push(a[i + 1], c);
        i += 4;
    };
    take_conv(9, Cin, C, r.conv_w[0], r.conv_b[0], &r.conv_k[0]);
    ci = 1;
    for (int k = 0; k < nb; ++k) {
        take_bn(C, r.bn_mean[bi], r.bn_inv[bi], r.bn_off[bi]); bi++;
        take_conv(9, C, C, r.conv_w[ci], r.conv_b[ci], &r.conv_k[ci], C == kRnC); ci++;
        take_bn(C, r.bn_mean[bi], r.bn_inv[bi], r.bn_off[bi]); bi++;
        take_conv(9, C, C, r.conv_w[ci], r.conv_b[ci], &r.conv_k[ci], C == kRnC); ci++;
    }
    take_bn(C, r.bn_mean[bi], r.bn_inv[bi], r.bn_off[bi]);  // final batch norm at index 2*nb
    auto take_lin = [&](int in, int outn, int& w_off, int& b_off) { w_off = push(a[i], static_cast<size_t>(in) * outn); b_off = push(a[i + 1], outn); i += 2; };
    take_conv(1, C, 2, r.hc_w[0], r.hc_b[0], nullptr); take_bn(2, r.hbn_mean[0], r.hbn_inv[0], r.hbn_off[0]); take_lin(r.P * 2, A, r.lin_w[0], r.lin_b[0]);
    take_conv(1, C, 2, r.hc_w[1], r.hc_b[1], nullptr); take_bn(2, r.hbn_mean[1], r.hbn_inv[1], r.hbn_off[1]); take_lin(r.P * 2, A, r.lin_w[1], r.lin_b[1]);
    take_conv(1, C, 1, r.hc_w[2], r.hc_b[2], nullptr); take_bn(1, r.hbn_mean[2], r.hbn_inv[2], r.hbn_off[2]);
    take_lin(r.P, C, r.lin_w[2], r.lin_b[2]); take_lin(C, 1, r.lin_w[3], r.lin_b[3]);
    take_conv(1, C, 1, r.hc_w[3], r.hc_b[3], nullptr); take_bn(1, r.hbn_mean[3], r.hbn_inv[3], r.hbn_off[3]);
    take_lin(r.P, C, r.lin_w[4], r.lin_b[4]); take_lin(C, 1, r.lin_w[5], r.lin_b[5]);
}

static int net_create_resnet_impl(eaz_net* n, int H, int W, int Cin, int C, int nb, int A, const float* const* a, int hash_bits) {
    std::vector<float> pk;
    resnet_pack(a, H, W, Cin, C, nb, A, n->rdev, pk);
    n->rparams_count = pk.size();
    HIP_OK(hipMalloc(&n->d_rparams, pk.size() * sizeof(float)));
    HIP_OK(hipMemcpy(n->d_rparams, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    n->rdev.p = n->d_rparams;
    n->dev.in_dim = H * W * Cin; n->dev.H = C; n->dev.A = A;
    n->dev.max_u = 1.0f; n->dev.nov_scale = 1.0f;  // resnet.py:63-64
    n->dev.hash_off = 0; n->dev.hash_len = H * W * Cin; n->dev.hash_bits = hash_bits;
    n->bitset_bytes = static_cast<size_t>(1) << (hash_bits - 3);
    if (n->bitset_bytes < 4) n->bitset_bytes = 4;
    HIP_OK(hipMalloc(&n->dev.bitset, n->bitset_bytes));
    HIP_OK(hipMemset(n->dev.bitset, 0, n->bitset_bytes));
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(resnet_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kRnLdsBytes));
    HIP_OK(hipMalloc(&n->d_kstep_storage, 8 * sizeof(unsigned long long)));
    HIP_OK(hipMemset(n->d_kstep_storage, 0, 8 * sizeof(unsigned long long)));
    return 0;
}

extern "C" int eaz_net_destroy(eaz_net* n);

extern "C" int eaz_net_create_resnet(eaz_ctx* ctx, int H, int W, int Cin, int C, int nb, int A, const float* const* a, int n_arrays,
                                     int hash_bits, eaz_net** out) {
    ARG_OK(ctx && out && a && H > 0 && W > 0 && A > 0 && A <= 64 && nb >= 0 && nb <= 16);
    if (C != kRnC) return fail("the gfx950 ResNet kernel is specialised for num_channels = %d (resnet.py:48), got %d", kRnC, C);
    if (Cin < 1 || Cin > 64 || (Cin & (Cin - 1)) != 0) return fail("input planes must be a power of two <= 64, got %d", Cin);
    if (n_arrays != 42 + 12 * nb) return fail("expected %d parameter arrays, got %d", 42 + 12 * nb, n_arrays);
    if ((H * W * Cin) % 4 != 0) return fail("hash input length %d is not a multiple of 4 (hashes.py:210)", H * W * Cin);
    if (H * W > 42) return fail("board of %d cells exceeds the kernel's 42-cell head buffers", H * W);
    ARG_OK(hash_bits >= 4 && hash_bits <= 32);
    HIP_OK(hipSetDevice(ctx->device));
    eaz_net* n = new eaz_net();  // value-initialised: every pointer null, so a failed creation can be handed to eaz_net_destroy
    n->ctx = ctx;
    n->kind = 1;
    const int rc = net_create_resnet_impl(n, H, W, Cin, C, nb, A, a, hash_bits);
    if (rc) { (void)eaz_net_destroy(n); return rc; }
    *out = n;
    return 0;
}

extern "C" int eaz_net_destroy(eaz_net* n) {
    if (!n) return 0;
    if (n->d_rparams) (void)hipFree(n->d_rparams);
    for (int l = 0; l < 12; ++l) { if (n->d_w[l]) (void)hipFree(n->d_w[l]); if (n->d_b[l]) (void)hipFree(n->d_b[l]); }
    if (n->dev.bitset) (void)hipFree(n->dev.bitset);
    if (n->d_ztab) (void)hipFree(n->d_ztab);
    if (n->d_hot_index) (void)hipFree(n->d_hot_index);
    if (n->d_sim_matrix) (void)hipFree(n->d_sim_matrix);
    if (n->d_kstep_storage) (void)hipFree(n->d_kstep_storage);
    for (int h = 0; h < 4; ++h) if (n->d_w1q[h]) (void)hipFree(n->d_w1q[h]);
    for (int h = 0; h < 4; ++h) if (n->d_w_raw[h]) (void)hipFree(n->d_w_raw[h]);
    if (n->d_in_perm) (void)hipFree(n->d_in_perm);
    delete n;
    return 0;
}

extern "C" int eaz_net_set_weights(eaz_net* n, const float* const* h_w, const float* const* h_b, void* stream) {
    ARG_OK(n && h_w && h_b);
    if (n->kind != 0) return fail("eaz_net_set_weights: FC networks only (recreate a ResNet with eaz_net_create_resnet)");
    for (int l = 0; l < 12; ++l) {
        int k, m;
        layer_dims(n->dev.in_dim, n->dev.H, n->dev.A, l, &k, &m);
        if ((l % 3) == 0 && n->dev.in_perm) {
            // an input order is set: the upload lands in the original-order copy, the live copy gets its rows in chain order
            HIP_OK(hipMemcpyAsync(n->d_w_raw[l / 3], h_w[l], sizeof(float) * k * m, hipMemcpyHostToDevice, S(stream)));
            fc_permute_w1_kernel<<<k, 256, 0, S(stream)>>>(n->d_w_raw[l / 3], n->dev.in_perm, n->d_w[l], m);
            LAUNCH_OK();
        } else {
            HIP_OK(hipMemcpyAsync(n->d_w[l], h_w[l], sizeof(float) * k * m, hipMemcpyHostToDevice, S(stream)));
        }
        HIP_OK(hipMemcpyAsync(n->d_b[l], h_b[l], sizeof(float) * m, hipMemcpyHostToDevice, S(stream)));
    }
    n->w_stamp += 1;  // derived device copies (d_w1q) are rebuilt by their next user
    return 0;
}

extern "C" int eaz_net_set_input_order(eaz_net* n, int n_seg, const int32_t* seg_start, const int32_t* seg_len, void* stream) {
    ARG_OK(n && n_seg >= 0 && (n_seg == 0 || (seg_start && seg_len)));
    if (n->kind != 0) return fail("eaz_net_set_input_order: FC networks only");
    HIP_OK(hipSetDevice(n->ctx->device));
    const int K = n->dev.in_dim, H = n->dev.H;
    std::vector<int32_t> perm;
    if (n_seg > 0) {
        std::vector<uint8_t> seen(static_cast<size_t>(K), 0);
        for (int s = 0; s < n_seg; ++s)
            for (int k = seg_start[s]; k < seg_start[s] + seg_len[s]; ++k) {
                if (k < 0 || k >= K || seen[k]) return fail("eaz_net_set_input_order: the segments are not a permutation of the %d inputs (input %d)", K, k);
                seen[k] = 1;
                perm.push_back(k);
            }
        if (static_cast<int>(perm.size()) != K) return fail("eaz_net_set_input_order: the segments cover %zu of the %d inputs", perm.size(), K);
        bool ident = true;
        for (int k = 0; k < K; ++k) ident = ident && perm[k] == k;
        if (ident) perm.clear();
    }
    hipStream_t st = S(stream);
    // the weights in their original order: kept beside the live copies from the first non-trivial order on
    for (int h = 0; h < 4; ++h) {
        if (!n->d_w_raw[h]) {
            if (perm.empty()) continue;  // ascending before and after: nothing to do
            HIP_OK(hipMalloc(&n->d_w_raw[h], sizeof(float) * K * H));
            HIP_OK(hipMemcpyAsync(n->d_w_raw[h], n->d_w[h * 3], sizeof(float) * K * H, hipMemcpyDeviceToDevice, st));  // no order was set: d_w is the original
        }
    }
    if (perm.empty()) {
        if (n->d_w_raw[0])
            for (int h = 0; h < 4; ++h) HIP_OK(hipMemcpyAsync(n->d_w[h * 3], n->d_w_raw[h], sizeof(float) * K * H, hipMemcpyDeviceToDevice, st));
        n->dev.in_perm = nullptr; n->dev.in_inv = nullptr;
        for (int h = 0; h < 4; ++h) n->dev.w1raw[h] = n->d_w[h * 3];
    } else {
        if (!n->d_in_perm) HIP_OK(hipMalloc(&n->d_in_perm, sizeof(int32_t) * 2 * K));
        std::vector<int32_t> both(static_cast<size_t>(2) * K);
        for (int kp = 0; kp < K; ++kp) { both[kp] = perm[kp]; both[K + perm[kp]] = kp; }
        HIP_OK(hipMemcpyAsync(n->d_in_perm, both.data(), sizeof(int32_t) * 2 * K, hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));  // `both` is a stack-lifetime staging buffer
        n->dev.in_perm = n->d_in_perm; n->dev.in_inv = n->d_in_perm + K;
        for (int h = 0; h < 4; ++h) {
            fc_permute_w1_kernel<<<K, 256, 0, st>>>(n->d_w_raw[h], n->dev.in_perm, n->d_w[h * 3], H);
            LAUNCH_OK();
            n->dev.w1raw[h] = n->d_w_raw[h];
        }
    }
    n->perm = perm;
    n->w_stamp += 1;
    return 0;
}

extern "C" int eaz_net_set_resnet_weights(eaz_net* n, const float* const* h_arrays, int n_arrays, void* stream) {
    ARG_OK(n && h_arrays);
    if (n->kind != 1) return fail("eaz_net_set_resnet_weights: residual networks only (FC: eaz_net_set_weights)");
    const ResNetDev& r0 = n->rdev;
    if (n_arrays != 42 + 12 * r0.nb) return fail("expected %d parameter arrays, got %d", 42 + 12 * r0.nb, n_arrays);
    ResNetDev r;
    std::vector<float> pk;
    resnet_pack(h_arrays, r0.H, r0.W, r0.Cin, kRnC, r0.nb, r0.A, r, pk);
    if (pk.size() != n->rparams_count) return fail("eaz_net_set_resnet_weights: packed size changed (%zu != %zu)", pk.size(), n->rparams_count);
    // same buffer, same offsets: kernels already captured into a hipGraph keep reading valid addresses and see the new weights
    HIP_OK(hipMemcpyAsync(n->d_rparams, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice, S(stream)));
    HIP_OK(hipStreamSynchronize(S(stream)));  // `pk` is a stack-lifetime staging buffer
    return 0;
}

extern "C" int eaz_net_set_bitset(eaz_net* n, const uint8_t* h_bitset, void* stream) {
    ARG_OK(n && h_bitset);
    HIP_OK(hipMemcpyAsync(n->dev.bitset, h_bitset, static_cast<size_t>(1) << (n->dev.hash_bits - 3), hipMemcpyHostToDevice, S(stream)));
    return 0;
}

extern "C" int eaz_net_get_bitset(eaz_net* n, uint8_t* h_bitset, void* stream) {
    ARG_OK(n && h_bitset);
    HIP_OK(hipMemcpyAsync(h_bitset, n->dev.bitset, static_cast<size_t>(1) << (n->dev.hash_bits - 3), hipMemcpyDeviceToHost, S(stream)));
    HIP_OK(hipStreamSynchronize(S(stream)));
    return 0;
}

static int hash_launch(eaz_net* n, int B, const uint8_t* obs, float* novelty, uint8_t* seen, uint32_t* idx, int mode, hipStream_t st) {
    if (B <= 0) return 0;
    if (n->dev.sim_matrix) simhash_kernel<<<cdiv(static_cast<long>(B) * 32, 256), 256, 0, st>>>(n->dev, obs, B, novelty, seen, idx, mode);
    else hash_kernel<<<cdiv(static_cast<long>(B) * 4, 256), 256, 0, st>>>(n->dev, obs, B, novelty, seen, idx, mode);
    LAUNCH_OK();
    return 0;
}

extern "C" int eaz_net_set_simhash(eaz_net* n, const float* h_random_matrix, void* stream) {
    ARG_OK(n && h_random_matrix);
    HIP_OK(hipSetDevice(n->ctx->device));
    const int len = n->dev.hash_len, bits = n->dev.hash_bits;
    if (!n->d_sim_matrix) HIP_OK(hipMalloc(&n->d_sim_matrix, sizeof(float) * len * bits));
    HIP_OK(hipMemcpyAsync(n->d_sim_matrix, h_random_matrix, sizeof(float) * len * bits, hipMemcpyHostToDevice, S(stream)));
    if (n->d_hot_index) {
        // index of every one-hot row under SimHash: product_j = fmaf(1, R[h][j], +0) = R[h][j]; rows outside the hashed slice give 0
        std::vector<uint32_t> hi(static_cast<size_t>(n->dev.in_dim), 0u);
        for (int h = 0; h < n->dev.in_dim; ++h) {
            const int hh = h - n->dev.hash_off;
            if (hh < 0 || hh >= len) continue;
            uint32_t v = 0;
            for (int j = 0; j < bits; ++j) v |= static_cast<uint32_t>(std::fmaf(1.0f, h_random_matrix[static_cast<size_t>(hh) * bits + j], 0.0f) < 0.0f) << j;
            hi[h] = v;
        }
        HIP_OK(hipMemcpyAsync(n->d_hot_index, hi.data(), hi.size() * sizeof(uint32_t), hipMemcpyHostToDevice, S(stream)));
    }
    HIP_OK(hipStreamSynchronize(S(stream)));
    n->dev.sim_matrix = n->d_sim_matrix;
    return 0;
}

// eaz_tune_set knobs (include/eaz.h); the environment variables EAZ_FC_TILE / EAZ_FUSED_LEAF give the initial values
static int g_fc_tile = [] { const char* e = std::getenv("EAZ_FC_TILE"); return e ? std::atoi(e) : 0; }();
static int g_fused_leaf = [] { const char* e = std::getenv("EAZ_FUSED_LEAF"); return e ? std::atoi(e) : 1; }();

static int g_tree_block = [] { const char* e = std::getenv("EAZ_TREE_BLOCK"); return e ? std::atoi(e) : 64; }();
static int g_tree_spread = [] { const char* e = std::getenv("EAZ_TREE_SPREAD"); return e ? std::atoi(e) : 0; }();  // 0 = automatic
static int g_persist = [] { const char* e = std::getenv("EAZ_PERSIST"); return e ? std::atoi(e) : 1; }();
static int g_shards = [] { const char* e = std::getenv("EAZ_SHARDS"); return e ? std::atoi(e) : 0; }();  // 0 = automatic
static int g_lazy_rows = [] { const char* e = std::getenv("EAZ_LAZY_ROWS"); return e ? std::atoi(e) : 1; }();  // 1 = automatic

extern "C" int eaz_tune_set(int knob, int value) {
    if (knob == EAZ_TUNE_TREE_BLOCK) {
        ARG_OK(value == 64 || value == 128 || value == 256);
        g_tree_block = value;
        return 0;
    }
    if (knob == EAZ_TUNE_TREE_SPREAD) {
        ARG_OK(value == 0 || value == 1 || value == 2 || value == 4 || value == 8);
        g_tree_spread = value;
        return 0;
    }
    if (knob == EAZ_TUNE_PERSIST) {
        ARG_OK(value == 0 || value == 1 || value == 2);  // 2: one launch per SEARCH only (Subleq-16: subleq16_search_kernel + the per-step kernels)
        g_persist = value;
        return 0;
    }
    if (knob == EAZ_TUNE_SHARDS) {
        ARG_OK(value == 0 || value == 1 || value == 2 || value == 4);  // 8 / 16 streams measured 328 k / 192 k env-steps/s at C3 (4: 559 k)
        g_shards = value;
        return 0;
    }
    if (knob == EAZ_TUNE_SUBLEQ_WAVE) {
        ARG_OK(value == 0 || value == 1);
        g_subleq_wave = value;
        return 0;
    }
    if (knob == EAZ_TUNE_LAZY_ROWS) {
        ARG_OK(value == 0 || value == 1 || value == 2);
        g_lazy_rows = value;
        return 0;
    }
    if (knob == EAZ_TUNE_FC_TILE) {
        ARG_OK(value == 0 || value == 1 || value == 16 || value == 64);
        g_fc_tile = value;
    } else if (knob == EAZ_TUNE_FUSED_LEAF) {
        g_fused_leaf = value != 0;
    } else {
        return fail("eaz_tune_set: unknown knob %d", knob);
    }
    return 0;
}

// forward of B one-hot observations given by index + their novelty lookup, ONE launch (fc16_forward_kernel<2>)
static int fc_onehot_launch(eaz_net* n, int B, const int32_t* hot, float* novelty, const NetOut& o, const HeadList& heads, hipStream_t st) {
    if (B <= 0 || heads.n <= 0) return 0;
    if (n->dev.in_perm) return fail("a layer-1 input order (eaz_net_set_input_order) is meaningless for one-hot observations: not supported with DeepSea");
    const int per_xcd = cdiv(cdiv(B, kF16M), 8);
    const int nb = 8 * per_xcd * heads.n;
    const FcIn in{nullptr, hot, novelty, n->dev.ztab};
    fc16_forward_kernel<2><<<nb + (novelty ? cdiv(B, 256) : 0), 256, f16_lds_bytes(n->dev.A), st>>>(n->dev, in, B, o, heads, per_xcd, nb, n->d_kstep_counter);
    LAUNCH_OK();
    return 0;
}

// `scratch`: [B, in_dim] bytes, needed when the network has a layer-1 input order (the rows are read through the permutation once, in front
// of whichever forward kernel runs: every kernel then computes the chain in that order)
static int fc_launch(eaz_net* n, int B, const uint8_t* obs, const NetOut& o, const HeadList& heads, hipStream_t st, uint8_t* scratch) {
    if (B <= 0 || heads.n <= 0) return 0;
    if (n->kind == 1) {  // residual tower: kRnSamples samples per workgroup, all heads in one launch
        resnet_forward_kernel<<<cdiv(B, kRnSamples), kRnThreads, kRnLdsBytes, st>>>(n->rdev, obs, B, o);
        LAUNCH_OK();
        return 0;
    }
    if (n->dev.in_perm) {
        if (!scratch) return fail("internal: a dense forward under a layer-1 input order needs a scratch row buffer");
        fc_permute_obs_kernel<<<cdiv(static_cast<long>(B) * n->dev.in_dim, 256), 256, 0, st>>>(obs, n->dev.in_perm, scratch, B, n->dev.in_dim);
        LAUNCH_OK();
        obs = scratch;
    }
    if (n->dev.H != kFcH) {  // any other linear_layer_size: plain fmaf chains
        fc_generic_forward_kernel<<<dim3(cdiv(B, kFgRows), heads.n), 256, fg_lds_bytes(n->dev.H), st>>>(n->dev, obs, B, o, heads);
        LAUNCH_OK();
        return 0;
    }
    // narrow heads on observations that fit an f32 LDS tile: the pipelined 16-row kernel (net_fc_pipe.hpp), one workgroup per tile, all
    // heads back to back
    if ((g_fc_tile == 0 || g_fc_tile == 1) && fp_eligible(n->dev.in_dim, n->dev.A)) {
        const int lds = fp_lds_bytes(n->dev.in_dim, n->dev.A);
        fc16p_forward_kernel<<<dim3(cdiv(B, kF16M), heads.n), 256, lds, st>>>(n->dev, obs, B, o, heads, n->d_kstep_counter);
        LAUNCH_OK();
        return 0;
    }
    // 16-row tiles while the whole launch is resident at once (<= 4 workgroups per CU), 64-row tiles for larger batches
    // (a quarter of the weight traffic per row)
    const int per_xcd = cdiv(cdiv(B, kF16M), 8);
    if (g_fc_tile == 16 || (g_fc_tile == 0 && static_cast<long>(8) * per_xcd * heads.n <= 4 * 256)) {
        const bool fast16 = (n->dev.in_dim % 16 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0);
        const FcIn in{obs, nullptr, nullptr, nullptr};
        const int nb = 8 * per_xcd * heads.n;
        if (fast16) fc16_forward_kernel<0><<<nb, 256, f16_lds_bytes(n->dev.A), st>>>(n->dev, in, B, o, heads, per_xcd, nb, n->d_kstep_counter);
        else fc16_forward_kernel<1><<<nb, 256, f16_lds_bytes(n->dev.A), st>>>(n->dev, in, B, o, heads, per_xcd, nb, n->d_kstep_counter);
        LAUNCH_OK();
        return 0;
    }
    const int n_tiles = cdiv(B, kFcM);
    int cmin = 8;
    for (int h = 0; h < heads.n; ++h) {
        int c = 0;
        for (int x = 0; x < 8; ++x) c += (x % heads.n) == h;
        cmin = c < cmin ? c : cmin;
    }
    const int grid = 8 * cdiv(n_tiles, cmin);
    const bool fast = (n->dev.in_dim % 16 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0);
    if (fast) fc_forward_kernel<true><<<grid, 256, kFcLdsBytes, st>>>(n->dev, obs, B, o, heads, n->d_kstep_counter);
    else fc_forward_kernel<false><<<grid, 256, kFcLdsBytes, st>>>(n->dev, obs, B, o, heads, n->d_kstep_counter);
    LAUNCH_OK();
    return 0;
}

extern "C" int eaz_net_forward(eaz_net* n, int B, const uint8_t* d_obs, float* d_exploit, float* d_explore, float* d_value,
                               float* d_ube, float* d_novelty, void* stream) {
    ARG_OK(n && B >= 0);
    if (B == 0) return 0;
    ARG_OK(d_obs != nullptr);
    float* nov = d_novelty;
    float* tmp = nullptr;
    if (d_ube && !nov) {  // u needs the novelty lookup even when the caller does not want it back
        HIP_OK(hipMallocAsync(reinterpret_cast<void**>(&tmp), sizeof(float) * B, S(stream)));
        nov = tmp;
    }
    if (nov) {
        int rc = hash_launch(n, B, d_obs, nov, nullptr, nullptr, 0, S(stream));
        if (rc) return rc;
    }
    HeadList hl{0, {0, 0, 0, 0}};
    if (d_value) hl.id[hl.n++] = 0;
    if (d_ube) hl.id[hl.n++] = 1;
    if (d_exploit) hl.id[hl.n++] = 2;
    if (d_explore) hl.id[hl.n++] = 3;
    NetOut o{d_exploit, d_explore, d_value, d_ube};
    uint8_t* scratch = nullptr;
    if (n->kind == 0 && n->dev.in_perm) HIP_OK(hipMallocAsync(reinterpret_cast<void**>(&scratch), static_cast<size_t>(B) * n->dev.in_dim, S(stream)));
    int rc = fc_launch(n, B, d_obs, o, hl, S(stream), scratch);
    if (scratch) HIP_OK(hipFreeAsync(scratch, S(stream)));
    if (rc) return rc;
    if (d_ube) {
        ube_combine_kernel<<<cdiv(B, 256), 256, 0, S(stream)>>>(d_ube, nov, n->dev.max_u, B);
        LAUNCH_OK();
    }
    if (tmp) HIP_OK(hipFreeAsync(tmp, S(stream)));
    return 0;
}

extern "C" int eaz_hash_lookup(eaz_net* n, int B, const uint8_t* d_obs, uint8_t* d_seen, uint32_t* d_indices, void* stream) {
    ARG_OK(n && B >= 0);
    if (B == 0) return 0;
    ARG_OK(d_obs != nullptr);
    return hash_launch(n, B, d_obs, nullptr, d_seen, d_indices, 0, S(stream));
}
extern "C" int eaz_hash_update(eaz_net* n, int B, const uint8_t* d_obs, void* stream) {
    ARG_OK(n && B >= 0);
    if (B == 0) return 0;
    ARG_OK(d_obs != nullptr);
    return hash_launch(n, B, d_obs, nullptr, nullptr, nullptr, 1, S(stream));
}

// ------------------------------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------------------------------
// mctx seq_halving.get_sequence_of_considered_visits [EXT] -- host-side integer table
static void considered_visits(int m, int S_, int32_t* out) {
    if (m <= 1) { for (int i = 0; i < S_; ++i) out[i] = i; return; }
    int log2max = 0;
    while ((1 << log2max) < m) log2max++;
    std::vector<int> visits(m, 0);
    int nc = m, len = 0;
    while (len < S_) {
        int extra = static_cast<int>(static_cast<double>(S_) / static_cast<double>(log2max * nc));
        if (extra < 1) extra = 1;
        for (int r = 0; r < extra; ++r) {
            for (int i = 0; i < nc && len < S_; ++i) out[len++] = visits[i];
            for (int i = 0; i < nc; ++i) visits[i] += 1;
        }
        nc = nc / 2 > 2 ? nc / 2 : 2;
    }
}

template <typename T>
static int dalloc(eaz_search* s, T** p, size_t count) {
    void* q = nullptr;
    HIP_OK(hipMalloc(&q, sizeof(T) * (count ? count : 1)));
    s->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return 0;
}
extern "C" int eaz_search_destroy(eaz_search* s);
// only used while eaz_search_create builds `s`: a failed allocation releases everything allocated so far
#define DALLOC(p, count)                              \
    do {                                              \
        int rc_ = dalloc(s, &(p), (count));           \
        if (rc_) { (void)eaz_search_destroy(s); return rc_; } \
    } while (0)

// dispatch on (lanes per tree, actions per lane)
#define TREE_DISPATCH(s, CALL)                                   \
    do {                                                         \
        switch ((s)->lpt * 8 + (s)->apl) {                       \
            case 2 * 8 + 1: { CALL(2, 1); } break;               \
            case 4 * 8 + 1: { CALL(4, 1); } break;               \
            case 8 * 8 + 1: { CALL(8, 1); } break;               \
            case 16 * 8 + 1: { CALL(16, 1); } break;             \
            case 32 * 8 + 1: { CALL(32, 1); } break;             \
            case 64 * 8 + 1: { CALL(64, 1); } break;             \
            case 64 * 8 + 2: { CALL(64, 2); } break;             \
            case 64 * 8 + 4: { CALL(64, 4); } break;             \
            default: return fail("unsupported action count");    \
        }                                                        \
    } while (0)

struct ProfScope {
    eaz_search* s;
    hipStream_t st;
    int idx;
    ProfScope(eaz_search* s_, int cls, hipStream_t st_) : s(s_), st(st_), idx(-1) {
        if (!s->prof_on) return;
        hipEvent_t a, b;
        if (s->ev_pool.size() >= 2) {
            a = s->ev_pool.back(); s->ev_pool.pop_back();
            b = s->ev_pool.back(); s->ev_pool.pop_back();
        } else {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        }
        s->spans.push_back(ProfSpan{a, b, cls});
        idx = static_cast<int>(s->spans.size()) - 1;
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(s->spans[idx].b, st);
    }
};

// a view of `p` over the trees [b0, b0 + Bs): every per-tree array pointer moved by b0 rows, own streams and events, nothing owned
static int make_shard(eaz_search* p, int b0, int Bs, eaz_search** out) {
    eaz_search* c = new eaz_search(*p);
    c->allocs.clear(); c->spans.clear(); c->ev_pool.clear(); c->shards.clear();
    c->step_graph = nullptr; c->graph_states = nullptr; c->graph_flags = -1; c->pending_expand = -1; c->prof_on = 0;
    c->persist_ok = 0; c->clear_pending = 0; c->d_replay_states = nullptr; c->d_stamps = nullptr;
    c->h_fit_flag = nullptr; c->d_fit_flag = nullptr;
    c->B = Bs;
    c->side = c->main = nullptr;
    c->ev_fork = c->ev_join = c->ev_in = c->ev_out = c->ev_shard_fork = c->ev_shard_join = nullptr;
    // No side stream of its own: the novelty lookup of a sub-batch runs in line on the sub-batch's stream (the sub-batches overlap each other
    // instead).  A side stream per sub-batch would be a SECOND level of fork / join inside the captured step (capturing stream -> sub-batch
    // stream -> lookup stream and back), and hipStreamEndCapture of the HIP runtime this library runs on -- PyTorch's bundled
    // libamdhip64.so 7.0.2, the one runtime of the process -- recurses without bound in hip::Stream::EndCapture() on
    // such a capture and overflows the stack (the round-2 host crash; rocgdb backtrace and the 60-line reproducer: DESIGN.md 4.3,
    // tools/graph_nested_fork.hip -- two sub-batches x one simulation are enough; /opt/rocm 7.2.0's runtime handles the same sequence).
    HIP_OK(hipStreamCreateWithFlags(&c->main, hipStreamNonBlocking));
    const long N = p->N, A = p->A, SB = p->tree.state_bytes, Sn = p->S, OD = p->env->dev.obs_dim;
    TreeDev& t = c->tree;
    t.B = Bs;
#define OFF(ptr, per_tree) ptr = ptr ? ptr + static_cast<long>(b0) * (per_tree) : ptr
    OFF(t.node_visits, N); OFF(t.parents, N); OFF(t.action_from_parent, N);
    OFF(t.raw_values, N); OFF(t.node_values, N); OFF(t.raw_var, N); OFF(t.node_var, N);
    OFF(t.children_index, N * A); OFF(t.children_visits, N * A); OFF(t.children_prior_logits, N * A); OFF(t.children_values, N * A);
    OFF(t.children_rewards, N * A); OFF(t.children_discounts, N * A); OFF(t.children_var, N * A); OFF(t.children_reward_var, N * A);
    OFF(t.beta, 1); OFF(t.root_invalid, A); OFF(t.root_gumbel, A); OFF(t.embeddings, N * SB);
    OFF(t.sim_parent, 1); OFF(t.sim_action, 1); OFF(t.sim_next, 1); OFF(t.path_node, Sn); OFF(t.path_action, Sn); OFF(t.path_len, 1);
    OFF(t.net_logits, A); OFF(t.net_value, 1); OFF(t.net_ube, 1); OFF(t.net_novelty, 1);
    OFF(t.next_reward, 1); OFF(t.next_term, 1); OFF(t.next_legal, A);
    OFF(t.fuse_reward, 1); OFF(t.fuse_term, 1); OFF(t.leaf_hot, 1);
    OFF(c->d_obs, OD); OFF(c->d_obs_perm, OD); OFF(c->d_novelty, 1); OFF(c->d_exploit, A); OFF(c->d_explore, A); OFF(c->d_v, 1); OFF(c->d_u, 1);
    OFF(c->d_next_reward, 1); OFF(c->d_next_term, 1); OFF(c->d_next_legal, A); OFF(c->d_leaf_hot, 1);
#undef OFF
    *out = c;
    return 0;
}

extern "C" int eaz_search_create(eaz_ctx* ctx, eaz_env* env, eaz_net* net, int B, const eaz_search_opts* opts, eaz_search** out) {
    ARG_OK(ctx && env && net && opts && out && B > 0);
    ARG_OK(opts->num_simulations >= 1 && opts->num_simulations <= 4096);
    if (net->dev.in_dim != env->dev.obs_dim) return fail("network input %d != observation size %d", net->dev.in_dim, env->dev.obs_dim);
    if (net->dev.A != env->dev.A) return fail("network actions %d != env actions %d", net->dev.A, env->dev.A);
    ARG_OK(opts->max_num_considered_actions >= 1 && opts->max_num_considered_actions <= 256);
    HIP_OK(hipSetDevice(ctx->device));
    eaz_search* s = new eaz_search();
    s->ctx = ctx; s->env = env; s->net = net; s->opts = *opts;
    s->B = B; s->S = opts->num_simulations; s->N = s->S + 1; s->A = env->dev.A;
    s->prof_on = 0; s->step_graph = nullptr; s->graph_states = nullptr; s->graph_flags = -1; s->pending_expand = -1;
    s->persist_ok = 0; s->persist_lds = 0; s->clear_pending = 0; s->sims_done = 0; s->d_stamps = nullptr;
    s->h_fit_flag = nullptr; s->d_fit_flag = nullptr; s->fit_epoch = 0;
    std::memset(&s->last_sp, 0, sizeof s->last_sp); s->d_replay_states = nullptr;
    s->sqp_lds = 0; s->d_sq_logits = nullptr; s->d_sq_l1init = nullptr; s->d_eval_wg = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&s->main, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming));
    s->graph_beta = 0.0f;
    HIP_OK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    const int A = s->A;
    int p2 = 2;
    while (p2 < A) p2 <<= 1;
    if (p2 <= 64) { s->lpt = p2; s->apl = 1; }
    else { s->lpt = 64; s->apl = p2 / 64; }
    TreeDev& t = s->tree;
    std::memset(&t, 0, sizeof t);
    t.B = B; t.N = s->N; t.A = A; t.S = s->S;
    const size_t bn = static_cast<size_t>(B) * s->N, bna = bn * A;
    DALLOC(t.node_visits, bn); DALLOC(t.parents, bn); DALLOC(t.action_from_parent, bn);
    DALLOC(t.raw_values, bn); DALLOC(t.node_values, bn); DALLOC(t.raw_var, bn); DALLOC(t.node_var, bn);
    DALLOC(t.children_index, bna); DALLOC(t.children_visits, bna);
    DALLOC(t.children_prior_logits, bna); DALLOC(t.children_values, bna); DALLOC(t.children_rewards, bna);
    DALLOC(t.children_discounts, bna); DALLOC(t.children_var, bna); DALLOC(t.children_reward_var, bna);
    DALLOC(t.beta, B); DALLOC(t.root_invalid, static_cast<size_t>(B) * A); DALLOC(t.root_gumbel, static_cast<size_t>(B) * A);
    t.state_bytes = env->dev.state_bytes;
    DALLOC(t.embeddings, bn * t.state_bytes);
    HIP_OK(hipMemset(t.embeddings, 0, bn * t.state_bytes));
    DALLOC(t.sim_parent, B); DALLOC(t.sim_action, B); DALLOC(t.sim_next, B);
    DALLOC(t.path_node, static_cast<size_t>(B) * s->S); DALLOC(t.path_action, static_cast<size_t>(B) * s->S); DALLOC(t.path_len, B);
    // read by mirror groups before their owner has written them (tree.hpp): never leftovers of a freed allocation
    HIP_OK(hipMemset(t.sim_parent, 0, sizeof(int32_t) * B)); HIP_OK(hipMemset(t.sim_action, 0, sizeof(int32_t) * B));
    HIP_OK(hipMemset(t.sim_next, 0, sizeof(int32_t) * B)); HIP_OK(hipMemset(t.path_len, 0, sizeof(int32_t) * B));
    HIP_OK(hipMemset(t.path_node, 0, sizeof(int32_t) * B * s->S)); HIP_OK(hipMemset(t.path_action, 0, sizeof(int32_t) * B * s->S));
    // sequential-halving table
    const int mc = opts->max_num_considered_actions;
    std::vector<int32_t> table(static_cast<size_t>(mc + 1) * s->S);
    for (int m = 0; m <= mc; ++m) considered_visits(m, s->S, table.data() + static_cast<size_t>(m) * s->S);
    DALLOC(s->d_table, table.size());
    HIP_OK(hipMemcpy(s->d_table, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    t.table = s->d_table;
    t.rescale = opts->rescale_values; t.use_mixed = opts->use_mixed_value; t.max_considered = mc;
    t.max_depth = opts->max_depth > 0 ? opts->max_depth : s->S;
    t.value_scale = opts->value_scale; t.maxvisit_init = opts->maxvisit_init;
    t.discount = opts->discount; t.two_player = opts->two_players_game;
    t.bonus_after = opts->epistemic_bonus_after_rescale != 0; t.var_backup = opts->epistemic_variance_backup != 0;
    t.lazy_rows = 0;  // set per real step by the self-play driver (LazyRows)
    // per-simulation workspace
    DALLOC(s->d_obs, static_cast<size_t>(B) * env->dev.obs_dim + 16);
    DALLOC(s->d_obs_perm, static_cast<size_t>(B) * env->dev.obs_dim + 16);
    DALLOC(s->d_novelty, B); DALLOC(s->d_exploit, static_cast<size_t>(B) * A); DALLOC(s->d_explore, static_cast<size_t>(B) * A);
    DALLOC(s->d_v, B); DALLOC(s->d_u, B); DALLOC(s->d_next_reward, B); DALLOC(s->d_next_term, B);
    DALLOC(s->d_next_legal, static_cast<size_t>(B) * A);
    DALLOC(s->d_root_logits, static_cast<size_t>(B) * A); DALLOC(s->d_root_beta, B); DALLOC(s->d_root_gumbel, static_cast<size_t>(B) * A);
    DALLOC(s->d_aw, static_cast<size_t>(B) * A); DALLOC(s->d_vprobs, static_cast<size_t>(B) * A);
    DALLOC(s->d_root_invalid, static_cast<size_t>(B) * A); DALLOC(s->d_legal, static_cast<size_t>(B) * A); DALLOC(s->d_action, B);
    t.net_logits = opts->exploration ? s->d_explore : s->d_exploit;  // context.py:132
    t.net_value = s->d_v; t.net_ube = s->d_u; t.net_novelty = s->d_novelty; t.max_u = net->dev.max_u;
    DALLOC(s->d_root_v, B); DALLOC(s->d_root_u, B); DALLOC(s->d_sum_value, B); DALLOC(s->d_sum_std, B);
    DALLOC(s->d_sum_qvar, static_cast<size_t>(B) * A); DALLOC(s->d_gumbel_stage, static_cast<size_t>(B) * A);
    DALLOC(s->d_step_keys, 4);
    DALLOC(s->d_step_out, 1);
    DALLOC(s->d_sum_q, static_cast<size_t>(B) * A); DALLOC(s->d_sum_vc, static_cast<size_t>(B) * A); DALLOC(s->d_next_v, B);
    DALLOC(s->d_term, B); DALLOC(s->d_term2, B); DALLOC(s->d_reward_tmp, B); DALLOC(s->d_alive, 1);
    t.next_reward = s->d_next_reward; t.next_term = s->d_next_term; t.next_legal = s->d_next_legal;
    DALLOC(s->d_leaf_hot, B);
    t.fenv = env->dev; t.fuse_reward = s->d_next_reward; t.fuse_term = s->d_next_term; t.leaf_hot = s->d_leaf_hot;
    t.fuse_env = 0;
    s->fuse_leaf = g_fused_leaf && env->dev.kind == EAZ_ENV_DEEPSEA && net->kind == 0 && net->dev.ztab != nullptr && net->dev.H == kFcH;
    if (s->fuse_leaf) HIP_OK(hipMemset(s->d_next_legal, 1, static_cast<size_t>(B) * A));  // legal_action_mask all true (deep_sea.py:19)
    {
        // persistent search: DeepSea + FC net, two actions, node ids / grid coordinates inside the packed LDS fields, and the 16 trees of a
        // workgroup + the forward tiles inside the CU's 160 KiB
        const PersistLds L = persist_lds(s->N, s->S, env->dev.G);
        s->persist_lds = L.total;
        s->persist_ok = s->fuse_leaf && A == kPsA && s->N < 32768 && env->dev.G < 4096 && net->dev.in_dim == env->dev.G * env->dev.G &&
                        mc >= 1 && L.total <= 160 * 1024;
        if (s->persist_ok)
            {
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(search_persist_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(search_persist_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
        }
        if (s->persist_ok) DALLOC(s->d_replay_states, static_cast<size_t>(B) * env->dev.state_bytes);
        if (s->persist_ok) DALLOC(s->d_eval_wg, static_cast<size_t>(1 + cdiv(B, kPsTrees)));
        // Subleq-16 one-launch search (search_subleq.hpp): two forward tile sets = 76 KB of dynamic LDS (eligibility is re-checked per run)
        if (env->dev.kind == EAZ_ENV_SUBLEQ && env->dev.W == 16 && net->kind == 0 && net->dev.H == kFcH && fp_eligible(net->dev.in_dim, net->dev.A)) {
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(subleq16_search_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       sq16_lds_bytes(net->dev.in_dim, net->dev.A, kSqHeadGroups)));
            // ... and the whole selfplay() call in one launch while the 16 trees + states of a workgroup fit beside the forward tiles
            // (node ids are bytes with 0xFF = none; eligibility of the network / hash class is re-checked per run)
            const QpLds QL = qp_lds(s->N, s->S);
            if (net->dev.in_dim == 240 && A == kQpA && s->N <= 250 && QL.total <= 160 * 1024) {
                s->sqp_lds = QL.total;
                HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(subleq16_selfplay_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, QL.total));
                HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(subleq16_selfplay_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, QL.total));
                DALLOC(s->d_sq_logits, static_cast<size_t>(cdiv(B, kQpTrees)) * kQpTrees * s->N * (2 * kQpA));
                DALLOC(s->d_sq_l1init, static_cast<size_t>(kQpL1Tab) + static_cast<size_t>(cdiv(B, kQpTrees)) * kQpTrees * 3 * kFcH);
                DALLOC(s->d_eval_wg, static_cast<size_t>(1 + cdiv(B, kQpTrees)));
                DALLOC(s->d_replay_states, static_cast<size_t>(B) * env->dev.state_bytes);
            }
        }
        // two half-batches on two streams where the env step is the interpreter (Subleq) and the batch is big enough to split
        s->ev_shard_fork = s->ev_shard_join = nullptr;
        int want = g_shards ? g_shards : (env->dev.kind == EAZ_ENV_SUBLEQ && B >= 2048 ? 4 : (env->dev.kind == EAZ_ENV_SUBLEQ && B >= 1024 ? 2 : 1));
        while (want > 1 && B / want < 64) want /= 2;
        if (want > 1) {
            HIP_OK(hipEventCreateWithFlags(&s->ev_shard_fork, hipEventDisableTiming));
            const int per = ((B / want) / 64) * 64;  // keeps the 16-byte alignment of every per-tree row
            for (int k = 0; k < want; ++k) {
                eaz_search* c = nullptr;
                const int b0 = k * per, Bs = k + 1 == want ? B - b0 : per;
                const int rc = make_shard(s, b0, Bs, &c);
                if (rc) { (void)eaz_search_destroy(s); return rc; }
                HIP_OK(hipEventCreateWithFlags(&c->ev_shard_join, hipEventDisableTiming));
                s->shards.push_back(c);
            }
        }
#ifdef EAZ_FC_STAMPS
        DALLOC(s->d_stamps, 4 * 256 * 32);  // search_subleq.hpp: [256][32] phase stamps + [256][8 waves][4]; search_subleq_persist.hpp: [256][8 waves][16]
        HIP_OK(hipMemset(s->d_stamps, 0, 4 * 256 * 32 * sizeof(unsigned long long)));
#endif
    }
    *out = s;
    return 0;
}

extern "C" int eaz_search_destroy(eaz_search* s) {
    if (!s) return 0;
    for (eaz_search* c : s->shards) (void)eaz_search_destroy(c);  // views: streams and events only
    s->shards.clear();
    if (s->ev_shard_fork && s->allocs.size()) (void)hipEventDestroy(s->ev_shard_fork);  // views hold a copy of the parent's handle
    if (s->ev_shard_join) (void)hipEventDestroy(s->ev_shard_join);
    for (void* p : s->allocs) (void)hipFree(p);
    for (auto& sp : s->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto& e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->step_graph) (void)hipGraphExecDestroy(s->step_graph);
    if (s->h_fit_flag) (void)hipHostFree(s->h_fit_flag);
    if (s->side) (void)hipStreamDestroy(s->side);
    if (s->main) (void)hipStreamDestroy(s->main);
    if (s->ev_in) (void)hipEventDestroy(s->ev_in);
    if (s->ev_out) (void)hipEventDestroy(s->ev_out);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    delete s;
    return 0;
}

static int tree_grid(const eaz_search* s) { return cdiv(static_cast<long>(s->B) * s->lpt, 256); }
static bool sqp_eligible(const eaz_search* s);
static int sqp_launch(eaz_search* s, const PersistSelfplay& sp, hipStream_t st);
static int sq16_roots_fit(eaz_search* s, const void* d_states, hipStream_t st, bool* fit);
static int ds_roots_fit(eaz_search* s, const void* d_states, hipStream_t st, bool* fit);

extern "C" int eaz_search_begin(eaz_search* s, const float* d_prior_logits, const float* d_value, const float* d_value_var,
                                const void* d_root_states, const float* d_beta, const uint8_t* d_invalid, const float* d_gumbel,
                                void* stream) {
    ARG_OK(s && d_prior_logits && d_value && d_value_var && d_root_states && d_beta && d_gumbel);
    hipStream_t st = S(stream);
    {
        ProfScope ps(s, EAZ_K_MISC, st);
        // The persistent kernel rewrites every tree array, so its searches skip the 24 MB clear; a caller that steps the search
        // simulation by simulation instead gets the (root-preserving) clear from ensure_cleared() before its first tree kernel.
        // (likewise the Subleq-16 one-launch kernel, search_subleq_persist.hpp)
        s->sqp_roots_fit = 0;
        s->ds_roots_fit = 0;
        if (s->persist_ok && g_persist) {
            bool fit = false;
            const int rc = ds_roots_fit(s, d_root_states, st, &fit);
            if (rc) return rc;
            s->ds_roots_fit = fit ? 1 : 0;
        }
        if (!(s->persist_ok && g_persist) && g_persist == 1 && sqp_eligible(s)) {
            bool fit = false;
            const int rc = sq16_roots_fit(s, d_root_states, st, &fit);
            if (rc) return rc;
            s->sqp_roots_fit = fit ? 1 : 0;
        }
        s->clear_pending = s->ds_roots_fit || s->sqp_roots_fit;
        if (!s->clear_pending) {
            const long bna = static_cast<long>(s->B) * s->N * s->A;
            tree_clear_kernel<<<cdiv(bna, 256) < 4096 ? cdiv(bna, 256) : 4096, 256, 0, st>>>(s->tree, 0);
            LAUNCH_OK();
        }
#define CALL_ROOT(L, P)                                                                                              \
    tree_root_kernel<L, P><<<tree_grid(s), 256, 0, st>>>(s->tree, d_prior_logits, d_value, d_value_var,                \
                                                         static_cast<const char*>(d_root_states), d_beta, d_invalid, d_gumbel)
        TREE_DISPATCH(s, CALL_ROOT);
#undef CALL_ROOT
        LAUNCH_OK();
    }
    s->pending_expand = -1;
    s->sims_done = 0;
    return 0;
}

static int ensure_cleared(eaz_search* s, hipStream_t st) {
    if (!s->clear_pending) return 0;
    ProfScope ps(s, EAZ_K_MISC, st);
    const long bna = static_cast<long>(s->B) * s->N * s->A;
    tree_clear_kernel<<<cdiv(bna, 256) < 4096 ? cdiv(bna, 256) : 4096, 256, 0, st>>>(s->tree, 1);
    LAUNCH_OK();
    s->clear_pending = 0;
    return 0;
}

static int persist_launch(eaz_search* s, hipStream_t st, const PersistSelfplay* sp = nullptr) {
    if (s->net->dev.in_perm) return fail("a layer-1 input order (eaz_net_set_input_order) is meaningless for one-hot observations: not supported with DeepSea");
    PersistArgs pa;
    if (sp) pa.sp = *sp;
    else std::memset(&pa.sp, 0, sizeof pa.sp);
    pa.t = s->tree;
    pa.t.fused_rf = 1;
    pa.net = s->net->dev;
    pa.env = s->env->dev;
    pa.head[0] = 0; pa.head[1] = 1; pa.head[2] = s->opts.exploration ? 3 : 2;  // context.py:132
    pa.amap_lds = s->env->dev.G * s->env->dev.G <= kPsAmapMax;
    pa.stamps = s->d_stamps;
    if (pa.sp.on == 2) search_persist_kernel<true><<<cdiv(s->B, kPsTrees), kPsThreads, s->persist_lds, st>>>(pa);
    else search_persist_kernel<false><<<cdiv(s->B, kPsTrees), kPsThreads, s->persist_lds, st>>>(pa);
    LAUNCH_OK();
    return 0;
}

// Subleq-16 + FC network whose novelty is the XXHash variant over the IO rows: every simulation of a search in one launch
// (search_subleq.hpp).  Decided per run: the hash class of a network can change after the search was created (eaz_net_set_simhash).
// layer-1 mode of the Subleq-16 one-launch kernel for this network's input order (search_subleq_persist.hpp sq16_l1_src): 0 = ascending,
// 1 = example_input | example_output | memory | input_after | output_after ("constants first"), -1 = any other order (kernel not eligible)
static int sq16_l1_mode_of(const eaz_net* n) {
    if (n->perm.empty()) return 0;
    if (n->dev.in_dim != 240) return -1;
    static const int seg[5][2] = {{80, 40}, {160, 40}, {0, 80}, {120, 40}, {200, 40}};
    int kp = 0;
    for (const auto& sg : seg)
        for (int k = sg[0]; k < sg[0] + sg[1]; ++k)
            if (n->perm[kp++] != k) return -1;
    return 1;
}
static bool sq16_persist_eligible(const eaz_search* s) {
    const EnvDev& ed = s->env->dev;
    if (!(ed.kind == EAZ_ENV_SUBLEQ && ed.W == 16 && ed.binary && ed.obs_dim == 240 && s->lpt == 16 && s->apl == 1 && s->net->kind == 0)) return false;
    const NetDev& nd = s->net->dev;
    // novelty: the XXHash variant over the IO rows (subleq_hash_only_io, config.py:30) or over the whole observation (fully_connected.py:84-89)
    const bool hash_ok = (nd.hash_off == 80 && nd.hash_len == 160) || (nd.hash_off == 0 && nd.hash_len == 240);
    return nd.H == kFcH && fp_eligible(nd.in_dim, nd.A) && nd.in_dim == 240 && !nd.sim_matrix && hash_ok;
}
static int sq16_persist_launch(eaz_search* s, int sim0, int n_sims, bool expand_first, bool expand_last, hipStream_t st) {
    Sq16SearchArgs a;
    a.t = s->tree;
    a.t.fused_rf = 1;
    a.t.fuse_env = 0;
    a.e = s->env->dev;
    a.net = s->net->dev;
    a.out = NetOut{s->opts.exploration ? nullptr : s->d_exploit, s->opts.exploration ? s->d_explore : nullptr, s->d_v, s->d_u};
    a.heads = HeadList{3, {0, 1, s->opts.exploration ? 3 : 2, 0}};  // context.py:132
    a.obs = s->d_obs;
    a.next_reward = s->d_next_reward; a.next_term = s->d_next_term; a.next_legal = s->d_next_legal;
    a.nf = NoveltyFuse{a.net.bitset, s->d_novelty, a.net.hash_bits, a.net.nov_scale};
    a.sim0 = sim0; a.n_sims = n_sims; a.expand_first = expand_first; a.expand_last = expand_last;
    a.stamps = s->d_stamps;
    subleq16_search_kernel<<<cdiv(s->B, 16), kSqThreads, sq16_lds_bytes(a.net.in_dim, a.net.A, kSqHeadGroups), st>>>(a);
    LAUNCH_OK();
    return 0;
}

// Subleq-16 fused self-play: T real steps in one launch (search_subleq_persist.hpp)
static bool sqp_eligible(const eaz_search* s) {
    if (!(s->sqp_lds > 0 && sq16_persist_eligible(s) && s->tree.max_depth >= 1 && sq16_l1_mode_of(s->net) >= 0)) return false;
    for (int i = 0; i < s->env->dev.n_tasks; ++i)  // the kernel keeps a tree's task as 16 bits (QpMisc::task); a root blob's own task field likewise
        if (s->env->dev.tasks[i] < -32768 || s->env->dev.tasks[i] > 32767) return false;
    return true;
}
// Do the blobs a call starts from fit the one-launch kernel's compact state (search_subleq_persist.hpp sq16_roots_fit_kernel)?  One small
// launch, a 4-byte copy and a stream synchronisation; while the stream is being captured the question cannot be answered and the answer
// is "no" (the per-simulation kernels keep whole blobs and are always right).
// roots a DeepSea one-launch call may start from (search_persist.hpp ds_roots_fit_kernel); inside a stream capture: "no" (the per-step kernels)
// one launch that stores this call's epoch into a pinned host word if any root does not fit, then a stream synchronisation: ~30 us per call
// (the first version -- memset, kernel, 4-byte copy into pageable memory, synchronisation -- cost 130 us, 0.4 % of a C2 launch)
static int roots_fit_flag(eaz_search* s) {
    if (s->h_fit_flag) return 0;
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&s->h_fit_flag), sizeof(int), hipHostMallocMapped));
    *s->h_fit_flag = 0;
    s->fit_epoch = 0;
    HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->d_fit_flag), s->h_fit_flag, 0));
    return 0;
}
static int ds_roots_fit(eaz_search* s, const void* d_states, hipStream_t st, bool* fit) {
    *fit = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;
    int rc = roots_fit_flag(s);
    if (rc) return rc;
    const int epoch = ++s->fit_epoch;
    ds_roots_fit_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(static_cast<const char*>(d_states), s->B, s->env->dev.G, s->d_fit_flag, epoch);
    LAUNCH_OK();
    HIP_OK(hipStreamSynchronize(st));
    *fit = *static_cast<volatile int*>(s->h_fit_flag) != epoch;
    return 0;
}
static int sq16_roots_fit(eaz_search* s, const void* d_states, hipStream_t st, bool* fit) {
    *fit = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;
    int rc = roots_fit_flag(s);
    if (rc) return rc;
    const int epoch = ++s->fit_epoch;
    sq16_roots_fit_kernel<<<cdiv(s->B, 256), 256, 0, st>>>(static_cast<const char*>(d_states), static_cast<long>(sizeof(eaz_subleq_state)), s->B, s->d_fit_flag, epoch);
    LAUNCH_OK();
    HIP_OK(hipStreamSynchronize(st));
    *fit = *static_cast<volatile int*>(s->h_fit_flag) != epoch;
    return 0;
}
// layer-1 weights in the one-launch kernel's packed input order (sq16_l1_src), rebuilt on the launch stream after every weight upload
static int sq16_pack_w1(eaz_net* n, hipStream_t st) {
    const int mode = sq16_l1_mode_of(n);
    if (mode < 0) return fail("internal: the Subleq-16 one-launch kernel does not support this layer-1 input order");
    if (n->dev.l1q_mode == mode && n->w1q_stamp == n->w_stamp) return 0;
    const int rows = 32 * sq16_l1_halves(0);  // the larger of the two modes
    for (int h = 0; h < 4; ++h)
        if (!n->d_w1q[h]) { HIP_OK(hipMalloc(&n->d_w1q[h], sizeof(float) * rows * kFcH)); n->dev.w1q[h] = n->d_w1q[h]; }
    sq16_pack_w1_kernel<<<dim3(32 * sq16_l1_halves(mode), 4), 256, 0, st>>>(n->dev, mode, n->d_w1q[0], n->d_w1q[1], n->d_w1q[2], n->d_w1q[3]);
    LAUNCH_OK();
    n->dev.l1q_mode = mode; n->dev.l1q_halves = sq16_l1_halves(mode);
    n->w1q_stamp = n->w_stamp;
    return 0;
}
static int sqp_launch(eaz_search* s, const PersistSelfplay& sp, hipStream_t st) {
    const int rc = sq16_pack_w1(s->net, st);
    if (rc) return rc;
    Sq16SelfplayArgs a;
    a.l1init = s->d_sq_l1init;
    a.sp = sp;
    a.t = s->tree;
    a.e = s->env->dev;
    a.net = s->net->dev;
    a.heads = HeadList{3, {0, 1, s->opts.exploration ? 3 : 2, 0}};  // context.py:132
    a.logits = s->d_sq_logits;
    a.stamps = s->d_stamps;
    if (a.net.l1q_mode == 1) {  // chain starts of the six task branches for the heads this search reads (search_subleq_persist.hpp qp_l1_chain)
        sq16_l1_table_kernel<<<dim3(6, 3), 256, 0, st>>>(a.net, a.heads, a.l1init);
        LAUNCH_OK();
    }
    if (a.sp.on == 2) subleq16_selfplay_kernel<true><<<cdiv(s->B, kQpTrees), kQpThreads, s->sqp_lds, st>>>(a);
    else subleq16_selfplay_kernel<false><<<cdiv(s->B, kQpTrees), kQpThreads, s->sqp_lds, st>>>(a);
    LAUNCH_OK();
    return 0;
}

static int tree_step_launch(eaz_search* s, int sim, int mode, hipStream_t st) {
    ProfScope ps(s, EAZ_K_TREE, st);
    // automatic: one-wavefront workgroups, and trees spread until the launch has at least one wavefront per CU
    int spread = g_tree_spread;
    if (spread <= 0) {
        spread = 1;
        while (spread < 8 && static_cast<long>(s->B) * s->lpt * spread < 256L * 64) spread *= 2;
    }
    const int tgrid = cdiv(static_cast<long>(s->B) * s->lpt * spread, g_tree_block);
#define CALL_STEP(L, P) tree_step_kernel<L, P><<<tgrid, g_tree_block, 0, st>>>(s->tree, sim, mode, spread)
    TREE_DISPATCH(s, CALL_STEP);
#undef CALL_STEP
    LAUNCH_OK();
    return 0;
}

// forward.apply on a batch of observations: the novelty lookup (hash_kernel, 4 lanes per sample, ~64 workgroups) and the
// MFMA forward (hundreds of workgroups) are independent, so the lookup is forked onto the search's side stream and joined
// before the consumer of both (tree expand / root preparation applies ube_combine).
static int infer_launch(eaz_search* s, const uint8_t* obs, float* novelty, const NetOut& o, const HeadList& hl, hipStream_t st,
                        bool novelty_done = false) {
    const int B = s->B;
    if (novelty_done) {  // the env-step kernel already looked the observations up
        ProfScope ps(s, EAZ_K_NET, st);
        return fc_launch(s->net, B, obs, o, hl, st, s->d_obs_perm);
    }
    hipStream_t hs = st;
    if (s->side) {
        HIP_OK(hipEventRecord(s->ev_fork, st));
        HIP_OK(hipStreamWaitEvent(s->side, s->ev_fork, 0));
        hs = s->side;
    }
    {
        ProfScope ps(s, EAZ_K_HASH, hs);
        int rc = hash_launch(s->net, B, obs, novelty, nullptr, nullptr, 0, hs);
        if (rc) return rc;
    }
    if (s->side) HIP_OK(hipEventRecord(s->ev_join, s->side));
    {
        ProfScope ps(s, EAZ_K_NET, st);
        int rc = fc_launch(s->net, B, obs, o, hl, st, s->d_obs_perm);
        if (rc) return rc;
    }
    if (s->side) HIP_OK(hipStreamWaitEvent(st, s->ev_join, 0));
    return 0;
}

// leaf gather + env.step + observation (context.py:125-127), novelty lookup, network forward (context.py:129-131)
static int recurrent_launch(eaz_search* s, hipStream_t st) {
    eaz_env* env = s->env;
    const int B = s->B;
    HeadList hl{3, {0, 1, s->opts.exploration ? 3 : 2, 0}};
    NetOut o{s->opts.exploration ? nullptr : s->d_exploit, s->opts.exploration ? s->d_explore : nullptr, s->d_v, s->d_u};
    if (s->tree.fuse_env) {
        // the tree kernel already stepped the environment and left the one-hot index of the new observation
        ProfScope ps(s, EAZ_K_NET, st);
        return fc_onehot_launch(s->net, B, s->d_leaf_hot, s->d_novelty, o, hl, st);
    }
    bool novelty_done = false;
    {
        ProfScope ps(s, EAZ_K_GATHER, st);
        if (env->dev.kind == EAZ_ENV_DEEPSEA) {
            leaf_gather_deepsea_kernel<<<cdiv(static_cast<long>(cdiv(B, kLgTrees)) * 64, 256), 256, 0, st>>>(
                env->dev, s->tree, s->d_obs, s->d_next_reward, s->d_next_term, s->d_next_legal);
            LAUNCH_OK();
        } else if (env->dev.kind == EAZ_ENV_CONNECT4) {
            StateRef src{s->tree.embeddings, s->tree.sim_parent, s->N};
            StateRef dst{s->tree.embeddings, s->tree.sim_next, s->N};
            const NetDev& nd = s->net->dev;
            if (g_fused_leaf) {
                // step + child state + observation row + novelty lookup in one launch (env_kernels.hpp connect4_leaf_kernel)
                novelty_done = s->net->kind == 1 && !nd.sim_matrix && nd.hash_off == 0 && nd.hash_len == 84 && nd.in_dim == 84;
                connect4_leaf_kernel<<<cdiv(static_cast<long>(B) * 4, 64), 64, 0, st>>>(
                    env->dev, src, dst, s->tree.sim_action, B, s->d_next_reward, s->d_next_term, s->d_next_legal, s->d_obs,
                    NoveltyFuse{nd.bitset, novelty_done ? s->d_novelty : nullptr, nd.hash_bits, nd.nov_scale});
                LAUNCH_OK();
            } else {
                connect4_step_kernel<<<cdiv(B, 256), 256, 0, st>>>(env->dev, src, dst, s->tree.sim_action, B, 0, Key{0, 0}, nullptr,
                                                                  s->d_next_reward, s->d_next_term, s->d_next_legal);
                LAUNCH_OK();
                obs_emit_kernel<<<cdiv(static_cast<long>(B) * 64, 256), 256, 0, st>>>(env->dev, dst, B, s->d_obs);
                LAUNCH_OK();
            }
        } else {
            StateRef src{s->tree.embeddings, s->tree.sim_parent, s->N};
            StateRef dst{s->tree.embeddings, s->tree.sim_next, s->N};
            // embedding slots 1.. are only ever written here and the allocation starts zeroed: bytes >= W of `memory` stay zero
            const int epw = subleq_envs_per_wg(B);
            // word_size 16 + binary encoding: the step kernel also writes the observation row (no obs_emit launch)
            const bool fused_obs = g_fused_leaf && env->dev.binary && ((env->dev.W == 16 && env->dev.obs_dim == 240) || (env->dev.W != 16 && g_subleq_wave));
            // ... and, for the XXHash variant over the IO rows (the FCNet default, fully_connected.py:85-87), the novelty lookup
            const NetDev& nd = s->net->dev;
            novelty_done = fused_obs && !nd.sim_matrix && nd.hash_off == 80 && nd.hash_len == 160 && nd.in_dim == 240;
            // other word sizes (wavefront-per-environment kernel): likewise when the hashed slice is exactly the four IO rows
            const bool wave_novelty = fused_obs && env->dev.W != 16 && s->net->kind == 0 && !nd.sim_matrix && nd.in_dim == env->dev.obs_dim &&
                                      nd.hash_off == env->dev.W * env->dev.obs_cols && nd.hash_len == 32 * env->dev.obs_cols;
            novelty_done = novelty_done || wave_novelty;
            if (env->dev.W == 16)
                subleq16_wave_kernel<<<B, 64, 0, st>>>(env->dev, src, dst, s->tree.sim_action, B, 0, Key{0, 0}, nullptr, nullptr, s->d_next_reward,
                                                       s->d_next_term, s->d_next_legal, 16, fused_obs ? s->d_obs : nullptr,
                                                       NoveltyFuse{nd.bitset, novelty_done ? s->d_novelty : nullptr, nd.hash_bits, nd.nov_scale});
            else if (g_subleq_wave)
                subleq_wave_kernel<<<B, 64, 0, st>>>(env->dev, src, dst, s->tree.sim_action, B, 0, Key{0, 0}, nullptr, nullptr, s->d_next_reward,
                                                     s->d_next_term, s->d_next_legal, fused_obs ? s->d_obs : nullptr,
                                                     NoveltyFuse{nd.bitset, wave_novelty ? s->d_novelty : nullptr, nd.hash_bits, nd.nov_scale});
            else
                subleq_step_kernel<<<cdiv(B, epw), 64, subleq_lds_bytes(env->dev.W), st>>>(env->dev, src, dst, s->tree.sim_action, B, 0, Key{0, 0},
                                                                                          nullptr, nullptr, s->d_next_reward, s->d_next_term,
                                                                                          s->d_next_legal, epw, (env->dev.W + 15) & ~15);
            LAUNCH_OK();
            if (!fused_obs) {
                obs_emit_kernel<<<cdiv(static_cast<long>(B) * 64, 256), 256, 0, st>>>(env->dev, dst, B, s->d_obs);
                LAUNCH_OK();
            }
        }
    }
    return infer_launch(s, s->d_obs, s->d_novelty, o, hl, st, novelty_done);
}

extern "C" int eaz_search_simulate(eaz_search* s, int sim, void* stream) {
    ARG_OK(s && sim >= 0 && sim < s->S);
    hipStream_t st = S(stream);
    int rc = ensure_cleared(s, st);
    if (rc) return rc;
    s->tree.fused_rf = 1;
    s->tree.fuse_env = s->fuse_leaf ? 1 + s->env->dev.kind : 0;
    int mode = kTreeSelect | (s->pending_expand >= 0 ? kTreeExpand : 0);
    const EnvDev& ed = s->env->dev;
    if (g_fused_leaf && ed.kind == EAZ_ENV_SUBLEQ && ed.W == 16 && s->lpt == 16 && s->apl == 1 && s->net->kind == 0) {
        // tree step + leaf step (+ observation row and, where the hashed slice is the IO rows, novelty) in one launch
        const NetDev& nd = s->net->dev;
        const bool obs_done = ed.binary && ed.obs_dim == 240;
        const bool novelty_done = obs_done && !nd.sim_matrix && nd.hash_off == 80 && nd.hash_len == 160 && nd.in_dim == 240;
        {
            ProfScope ps(s, EAZ_K_TREE, st);
            subleq16_tree_leaf_kernel<<<s->B, 64, 0, st>>>(s->tree, sim, mode, ed, s->d_next_reward, s->d_next_term, s->d_next_legal,
                                                          obs_done ? s->d_obs : nullptr,
                                                          NoveltyFuse{nd.bitset, novelty_done ? s->d_novelty : nullptr, nd.hash_bits, nd.nov_scale});
            LAUNCH_OK();
        }
        if (!obs_done) {
            ProfScope ps(s, EAZ_K_GATHER, st);
            StateRef dst{s->tree.embeddings, s->tree.sim_next, s->N};
            obs_emit_kernel<<<cdiv(static_cast<long>(s->B) * 64, 256), 256, 0, st>>>(ed, dst, s->B, s->d_obs);
            LAUNCH_OK();
        }
        HeadList hl{3, {0, 1, s->opts.exploration ? 3 : 2, 0}};
        NetOut o{s->opts.exploration ? nullptr : s->d_exploit, s->opts.exploration ? s->d_explore : nullptr, s->d_v, s->d_u};
        rc = infer_launch(s, s->d_obs, s->d_novelty, o, hl, st, novelty_done);
        if (rc) return rc;
        s->pending_expand = sim;
        s->sims_done += 1;
        return 0;
    }
    static const int wide_tree_leaf = [] { const char* e = std::getenv("EAZ_SUBLEQ_TREE_LEAF"); return e ? std::atoi(e) : 1; }();
    if (wide_tree_leaf && g_fused_leaf && g_subleq_wave && ed.kind == EAZ_ENV_SUBLEQ && ed.W != 16 && ed.binary && s->lpt == 64 && s->net->kind == 0 &&
        (s->apl == 1 || s->apl == 2 || s->apl == 4)) {
        // word sizes 64..256 (256: the reference's code default): tree step + leaf step + observation row (+ novelty over the IO rows) in one
        // launch of one wavefront per tree (search_kernels.hpp subleq_tree_leaf_kernel), then the forward
        const NetDev& nd = s->net->dev;
        const bool novelty_done = !nd.sim_matrix && nd.in_dim == ed.obs_dim && nd.hash_off == ed.W * ed.obs_cols && nd.hash_len == 32 * ed.obs_cols;
        {
            ProfScope ps(s, EAZ_K_TREE, st);
            SqTreeLeafArgs ta;
            ta.t = s->tree; ta.sim = sim; ta.mode = mode; ta.e = ed;
            ta.next_reward = s->d_next_reward; ta.next_term = s->d_next_term; ta.next_legal = s->d_next_legal; ta.obs_out = s->d_obs;
            ta.nf = NoveltyFuse{nd.bitset, novelty_done ? s->d_novelty : nullptr, nd.hash_bits, nd.nov_scale};
            if (s->apl == 1) subleq_tree_leaf_kernel<1><<<s->B, 64, 0, st>>>(ta);
            else if (s->apl == 2) subleq_tree_leaf_kernel<2><<<s->B, 64, 0, st>>>(ta);  // word sizes 65..128
            else subleq_tree_leaf_kernel<4><<<s->B, 64, 0, st>>>(ta);
            LAUNCH_OK();
        }
        HeadList hl{3, {0, 1, s->opts.exploration ? 3 : 2, 0}};
        NetOut o{s->opts.exploration ? nullptr : s->d_exploit, s->opts.exploration ? s->d_explore : nullptr, s->d_v, s->d_u};
        rc = infer_launch(s, s->d_obs, s->d_novelty, o, hl, st, novelty_done);
        if (rc) return rc;
        s->pending_expand = sim;
        s->sims_done += 1;
        return 0;
    }
    rc = tree_step_launch(s, sim, mode, st);
    if (rc) return rc;
    rc = recurrent_launch(s, st);
    if (rc) return rc;
    s->pending_expand = sim;
    s->sims_done += 1;
    return 0;
}

static int flush_expand(eaz_search* s, hipStream_t st) {
    if (s->pending_expand < 0) return 0;
    int rc = tree_step_launch(s, s->pending_expand + 1, kTreeExpand, st);
    s->pending_expand = -1;
    return rc;
}

extern "C" int eaz_search_debug_poison(eaz_search* s, int what, int byte_value, void* stream) {
    ARG_OK(s != nullptr && byte_value >= 0 && byte_value <= 255 && (what == 0 || what == 1));
    hipStream_t st = S(stream);
    const TreeDev& t = s->tree;
    const size_t B = static_cast<size_t>(s->B), BS = B * s->S;
    if (what == 1) {  // the eight [B,N,A] child arrays: what lazy rows (TreeDev::lazy_rows) must never depend on
        const size_t bytes = sizeof(float) * B * s->N * s->A;
        void* rows[8] = {t.children_index, t.children_visits, t.children_prior_logits, t.children_values,
                         t.children_rewards, t.children_discounts, t.children_var, t.children_reward_var};
        for (void* r : rows) HIP_OK(hipMemsetAsync(r, byte_value, bytes, st));
        return 0;
    }
    HIP_OK(hipMemsetAsync(t.sim_parent, byte_value, sizeof(int32_t) * B, st)); HIP_OK(hipMemsetAsync(t.sim_action, byte_value, sizeof(int32_t) * B, st));
    HIP_OK(hipMemsetAsync(t.sim_next, byte_value, sizeof(int32_t) * B, st)); HIP_OK(hipMemsetAsync(t.path_len, byte_value, sizeof(int32_t) * B, st));
    HIP_OK(hipMemsetAsync(t.path_node, byte_value, sizeof(int32_t) * BS, st)); HIP_OK(hipMemsetAsync(t.path_action, byte_value, sizeof(int32_t) * BS, st));
    return 0;
}

extern "C" int eaz_search_launch_info(const eaz_search* s, int* one_launch, int* lds_bytes) {
    ARG_OK(s != nullptr);
    const bool ds = s->persist_ok && g_persist;
    const bool sq = !ds && g_persist == 1 && sqp_eligible(s);
    if (one_launch) *one_launch = (ds || sq) ? 1 : 0;
    if (lds_bytes) *lds_bytes = ds ? s->persist_lds : (sq ? s->sqp_lds : 0);
    return 0;
}

extern "C" int eaz_search_run(eaz_search* s, void* stream) {
    ARG_OK(s != nullptr);
    if (s->persist_ok && g_persist && s->clear_pending && s->ds_roots_fit && s->sims_done == 0 && s->pending_expand < 0) {
        // all S simulations of every tree in one launch, trees resident in LDS (search_persist.hpp)
        ProfScope ps(s, EAZ_K_SEARCH, S(stream));
        int rc = persist_launch(s, S(stream));
        if (rc) return rc;
        s->clear_pending = 0;
        s->sims_done = s->S;
        return 0;
    }
    if (g_persist == 1 && s->clear_pending && s->sims_done == 0 && s->pending_expand < 0 && s->sqp_roots_fit && sqp_eligible(s)) {
        // Subleq-16: all S simulations in one launch on LDS node-record trees, the whole tree written back at the end
        // (search_subleq_persist.hpp, search-only mode)
        ProfScope ps(s, EAZ_K_SEARCH, S(stream));
        PersistSelfplay sp;
        std::memset(&sp, 0, sizeof sp);
        int rc = sqp_launch(s, sp, S(stream));
        if (rc) return rc;
        s->clear_pending = 0;
        s->sims_done = s->S;
        return 0;
    }
    if (g_persist && s->sims_done == 0 && s->pending_expand < 0 && sq16_persist_eligible(s)) {
        // Subleq-16 with global-memory trees (EAZ_TUNE_PERSIST = 2, or trees beyond the LDS limit): all S simulations in one launch,
        // 16 trees per workgroup (search_subleq.hpp)
        hipStream_t st = S(stream);
        int rc = ensure_cleared(s, st);
        if (rc) return rc;
        ProfScope ps(s, EAZ_K_SEARCH, st);
        rc = sq16_persist_launch(s, 0, s->S, false, true, st);
        if (rc) return rc;
        s->sims_done = s->S;
        return 0;
    }
    if (s->shards.size() > 1 && !s->prof_on && s->pending_expand < 0 && s->sims_done == 0) {
        // sub-batches: shard 0 on the caller's stream, the others on their own streams (forked / joined with events: capturable); trees are
        // independent, so the results are those of the unsharded loop
        hipStream_t st = S(stream);
        int rc = ensure_cleared(s, st);
        if (rc) return rc;
        const size_t K = s->shards.size();
        for (eaz_search* c : s->shards) { c->opts = s->opts; c->tree.fused_rf = 1; c->pending_expand = -1; c->fuse_leaf = s->fuse_leaf; }
        HIP_OK(hipEventRecord(s->ev_shard_fork, st));
        for (size_t k = 1; k < K; ++k) HIP_OK(hipStreamWaitEvent(s->shards[k]->main, s->ev_shard_fork, 0));
        for (int sim = 0; sim < s->S; ++sim)
            for (size_t k = 0; k < K; ++k) {
                rc = eaz_search_simulate(s->shards[k], sim, k == 0 ? st : s->shards[k]->main);
                if (rc) return rc;
            }
        for (size_t k = 0; k < K; ++k) {
            rc = flush_expand(s->shards[k], k == 0 ? st : s->shards[k]->main);
            if (rc) return rc;
        }
        for (size_t k = 1; k < K; ++k) {
            HIP_OK(hipEventRecord(s->shards[k]->ev_shard_join, s->shards[k]->main));
            HIP_OK(hipStreamWaitEvent(st, s->shards[k]->ev_shard_join, 0));
        }
        s->sims_done = s->S;
        return 0;
    }
    for (int sim = 0; sim < s->S; ++sim) {
        int rc = eaz_search_simulate(s, sim, stream);
        if (rc) return rc;
    }
    return flush_expand(s, S(stream));
}

extern "C" int eaz_search_finish(eaz_search* s, int32_t* d_action, float* d_action_weights, void* stream) {
    ARG_OK(s != nullptr);
    hipStream_t st = S(stream);
    int rc = ensure_cleared(s, st);
    if (rc) return rc;
    rc = flush_expand(s, st);
    if (rc) return rc;
    ProfScope ps(s, EAZ_K_MISC, st);
#define CALL_FIN(L, P) tree_finish_kernel<L, P><<<tree_grid(s), 256, 0, st>>>(s->tree, d_action, d_action_weights)
    TREE_DISPATCH(s, CALL_FIN);
#undef CALL_FIN
    LAUNCH_OK();
    return 0;
}

extern "C" int eaz_search_summary(eaz_search* s, float* d_value, float* d_value_std, float* d_qvalues, float* d_qvalues_var,
                                  float* d_visit_counts, float* d_visit_probs, void* stream) {
    ARG_OK(s != nullptr);
    hipStream_t st = S(stream);
    int rc = ensure_cleared(s, st);
    if (rc) return rc;
    rc = flush_expand(s, st);
    if (rc) return rc;
    ProfScope ps(s, EAZ_K_MISC, st);
    tree_summary_kernel<<<cdiv(static_cast<long>(s->B) * s->A, 256), 256, 0, st>>>(s->tree, d_value, d_value_std, d_qvalues,
                                                                                   d_qvalues_var, d_visit_counts, d_visit_probs);
    LAUNCH_OK();
    return 0;
}

extern "C" int eaz_search_select(eaz_search* s, int sim, int32_t* d_parent, int32_t* d_action, int32_t* d_next, void* stream) {
    ARG_OK(s && sim >= 0 && sim < s->S);
    hipStream_t st = S(stream);
    if (s->pending_expand >= 0) return fail("eaz_search_select: simulation %d has not been expanded yet", s->pending_expand);
    int rc = ensure_cleared(s, st);
    if (rc) return rc;
    s->tree.fuse_env = 0;  // the caller's recurrent_fn steps the environment
    rc = tree_step_launch(s, sim, kTreeSelect, st);
    if (rc) return rc;
    const size_t n = sizeof(int32_t) * s->B;
    if (d_parent) HIP_OK(hipMemcpyAsync(d_parent, s->tree.sim_parent, n, hipMemcpyDeviceToDevice, st));
    if (d_action) HIP_OK(hipMemcpyAsync(d_action, s->tree.sim_action, n, hipMemcpyDeviceToDevice, st));
    if (d_next) HIP_OK(hipMemcpyAsync(d_next, s->tree.sim_next, n, hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int eaz_search_expand(eaz_search* s, int sim, const float* d_reward, const float* d_reward_var, const float* d_discount,
                                 const float* d_prior_logits, const float* d_value, const float* d_value_var, void* stream) {
    ARG_OK(s && d_reward && d_reward_var && d_discount && d_prior_logits && d_value && d_value_var);
    TreeDev& t = s->tree;
    {
        int rc0 = ensure_cleared(s, S(stream));
        if (rc0) return rc0;
    }
    t.fused_rf = 0;
    t.rf_reward = d_reward; t.rf_reward_var = d_reward_var; t.rf_discount = d_discount;
    t.rf_logits = d_prior_logits; t.rf_value = d_value; t.rf_value_var = d_value_var;
    int rc = tree_step_launch(s, sim + 1, kTreeExpand, S(stream));
    t.fused_rf = 1;
    return rc;
}

extern "C" int eaz_search_tree_view(eaz_search* s, int id, void** d_ptr, int64_t shape[3], int* elem_bytes) {
    ARG_OK(s && d_ptr && shape && elem_bytes);
    TreeDev& t = s->tree;
    const int64_t B = s->B, N = s->N, A = s->A;
    void* p = nullptr;
    int64_t sh[3] = {B, N, 0};
    int eb = 4;
    switch (id) {
        case EAZ_F_NODE_VISITS: p = t.node_visits; break;
        case EAZ_F_RAW_VALUES: p = t.raw_values; break;
        case EAZ_F_NODE_VALUES: p = t.node_values; break;
        case EAZ_F_PARENTS: p = t.parents; break;
        case EAZ_F_ACTION_FROM_PARENT: p = t.action_from_parent; break;
        case EAZ_F_RAW_VALUES_VAR: p = t.raw_var; break;
        case EAZ_F_NODE_VALUES_VAR: p = t.node_var; break;
        case EAZ_F_CHILDREN_INDEX: p = t.children_index; sh[2] = A; break;
        case EAZ_F_CHILDREN_PRIOR_LOGITS: p = t.children_prior_logits; sh[2] = A; break;
        case EAZ_F_CHILDREN_VALUES: p = t.children_values; sh[2] = A; break;
        case EAZ_F_CHILDREN_VISITS: p = t.children_visits; sh[2] = A; break;
        case EAZ_F_CHILDREN_REWARDS: p = t.children_rewards; sh[2] = A; break;
        case EAZ_F_CHILDREN_DISCOUNTS: p = t.children_discounts; sh[2] = A; break;
        case EAZ_F_CHILDREN_VALUES_VAR: p = t.children_var; sh[2] = A; break;
        case EAZ_F_CHILDREN_REWARDS_VAR: p = t.children_reward_var; sh[2] = A; break;
        case EAZ_F_EMBEDDINGS: p = t.embeddings; sh[2] = t.state_bytes; eb = 1; break;
        case EAZ_F_ROOT_GUMBEL: p = t.root_gumbel; sh[1] = A; break;
        case EAZ_F_BETA: p = t.beta; sh[1] = 0; break;
        default: return fail("unknown tree field id %d", id);
    }
    *d_ptr = p;
    shape[0] = sh[0]; shape[1] = sh[1]; shape[2] = sh[2];
    *elem_bytes = eb;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// self-play driver (src/selfplay.py:78-149)
// ------------------------------------------------------------------------------------------------------------------
// destination slots of SelfplayOutput[t] (selfplay.py:136-143); any pointer may be null
struct StepOut {
    float *value_pred, *ube_pred, *root_value, *root_std, *qvar;
    int32_t* action;
    char* states;
};

__global__ void set_step_keys_kernel(Key* keys, Key k0, Key k1, Key k2, Key k3, StepOut* slot, StepOut o) {
    keys[0] = k0; keys[1] = k1; keys[2] = k2; keys[3] = k3;
    if (slot) *slot = o;
}

// SelfplayOutput[t]: one launch (the last node of the captured step) instead of seven device-to-device copies per real step;
// the destinations are read from the device-resident slot that set_step_keys_kernel refreshed for this step
__global__ void __launch_bounds__(256) selfplay_emit_kernel(int B, int A, int state_bytes, const StepOut* slot, const float* root_v,
                                                            const float* root_u, const float* sum_value, const float* sum_std,
                                                            const float* sum_qvar, const int32_t* action, const char* states) {
    const StepOut o = *slot;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x, i0 = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (long i = i0; i < B; i += stride) {
        if (o.value_pred) o.value_pred[i] = root_v[i];
        if (o.ube_pred) o.ube_pred[i] = root_u[i];
        if (o.root_value) o.root_value[i] = sum_value[i];
        if (o.root_std) o.root_std[i] = sum_std[i];
        if (o.action) o.action[i] = action[i];
    }
    if (o.qvar)
        for (long i = i0; i < static_cast<long>(B) * A; i += stride) o.qvar[i] = sum_qvar[i];
    if (o.states) {
        const long n16 = static_cast<long>(B) * state_bytes / 16;  // every state blob is a multiple of 16 bytes (include/eaz.h)
        for (long i = i0; i < n16; i += stride) reinterpret_cast<uint4*>(o.states)[i] = reinterpret_cast<const uint4*>(states)[i];
    }
}

enum : int { kFlagUniform = 1, kFlagSampleVisits = 2, kFlagSampleImproved = 4, kFlagExplicitGumbel = 8 };

// One real step of step_fn (selfplay.py:86-143) on fixed addresses: everything that changes from step to step (keys, explicit
// root noise, output slots) is read from / written to staging buffers, so the launch sequence can be captured ONCE into a
// hipGraph and replayed (S+1 network forwards, S+1 tree kernels, ...: 2S+13 launches per real step on the fused DeepSea path,
// 3S-4S+14 otherwise).
// Lazy child rows (tree.hpp TreeDev::lazy_rows) for the searches of the per-step self-play driver, whose trees nobody reads back (include/eaz.h:
// after eaz_selfplay_run the tree arrays are scratch): Subleq word_size 256 clears 2.2 GB of [B,N,A] arrays per real step at B = 4096 otherwise
// (0.39 ms of 8.5).  Off for narrow trees, where the clear is a few MB and the expand's extra row store is not free.
struct LazyRows {
    eaz_search* s;
    explicit LazyRows(eaz_search* q) : s(q) { set(g_lazy_rows == 2 || (g_lazy_rows == 1 && q->A >= 64)); }
    ~LazyRows() { set(false); }
    void set(bool on) const {
        s->tree.lazy_rows = on;
        for (eaz_search* c : s->shards) c->tree.lazy_rows = on;
    }
};

static int selfplay_step_body(eaz_search* s, const eaz_selfplay_cfg* cfg, int flags, void* d_states, hipStream_t st) {
    eaz_env* env = s->env;
    const int B = s->B, A = s->A;
    int rc;
    const LazyRows lazy(s);  // from here to the summary below: every tree kernel of this real step
    // root inference (selfplay.py:89-91): observation -> {novelty lookup || value / ube / policy heads}
    {
        ProfScope ps(s, EAZ_K_MISC, st);
        rc = obs_launch(env, B, d_states, s->d_obs, st);
        if (rc) return rc;
    }
    const bool explore = cfg->directed_exploration != 0;  // selfplay.py:93-95
    {
        HeadList hl{3, {0, 1, explore ? 3 : 2, 0}};
        NetOut o{explore ? nullptr : s->d_root_logits, explore ? s->d_root_logits : nullptr, s->d_root_v, s->d_root_u};
        rc = infer_launch(s, s->d_obs, s->d_novelty, o, hl, st);
        if (rc) return rc;
    }
    {
        ProfScope ps(s, EAZ_K_MISC, st);
        env_fields_kernel<<<cdiv(B, 256), 256, 0, st>>>(env->dev, static_cast<const char*>(d_states), B, nullptr, nullptr, nullptr,
                                                       nullptr, s->d_legal, nullptr);
        LAUNCH_OK();
        const float sbeta = explore ? cfg->exploration_beta : 0.0f;  // selfplay.py:92
        root_prep_kernel<<<cdiv(static_cast<long>(B) * A, 256), 256, 0, st>>>(
            B, A, s->d_root_logits, (flags & kFlagUniform) != 0, sbeta, s->d_legal,
            (flags & kFlagExplicitGumbel) ? s->d_gumbel_stage : nullptr, s->d_step_keys, s->opts.gumbel_scale, s->d_root_logits,
            s->d_root_beta, s->d_root_invalid, s->d_root_gumbel, s->d_root_u, s->d_novelty, s->net->dev.max_u, 0);
        LAUNCH_OK();
    }
    rc = eaz_search_begin(s, s->d_root_logits, s->d_root_v, s->d_root_u, d_states, s->d_root_beta, s->d_root_invalid, s->d_root_gumbel, st);
    if (rc) return rc;
    rc = eaz_search_run(s, st);
    if (rc) return rc;
    rc = eaz_search_finish(s, s->d_action, s->d_aw, st);
    if (rc) return rc;
    rc = eaz_search_summary(s, s->d_sum_value, s->d_sum_std, nullptr, s->d_sum_qvar, nullptr, s->d_vprobs, st);
    if (rc) return rc;
    {
        ProfScope ps(s, EAZ_K_MISC, st);
        if (flags & (kFlagSampleVisits | kFlagSampleImproved)) {
            choose_action_kernel<<<cdiv(B, 256), 256, 0, st>>>(B, A, s->d_action, s->d_vprobs, s->d_aw, (flags & kFlagSampleVisits) != 0,
                                                              (flags & kFlagSampleImproved) != 0, s->d_step_keys);
            LAUNCH_OK();
        }
        // real env step with auto-reset (selfplay.py:135), in place
        rc = env_step_launch(env, B, d_states, s->d_action, Key{0, 0}, 1, d_states, nullptr, st, s->d_step_keys + 1);
        if (rc) return rc;
        const long work = static_cast<long>(B) * (env->dev.state_bytes / 16 > A ? env->dev.state_bytes / 16 : A);
        selfplay_emit_kernel<<<cdiv(work, 256) < 2048 ? cdiv(work, 256) : 2048, 256, 0, st>>>(
            B, A, env->dev.state_bytes, s->d_step_out, s->d_root_v, s->d_root_u, s->d_sum_value, s->d_sum_std, s->d_sum_qvar, s->d_action,
            static_cast<const char*>(d_states));
        LAUNCH_OK();
    }
    return 0;
}

static int selfplay_capture(eaz_search* s, const eaz_selfplay_cfg* cfg, int flags, void* d_states, hipStream_t st) {
    if (s->step_graph && s->graph_states == d_states && s->graph_flags == flags) return 0;
    if (s->step_graph) { (void)hipGraphExecDestroy(s->step_graph); s->step_graph = nullptr; }
    hipGraph_t graph = nullptr;
    HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = selfplay_step_body(s, cfg, flags, d_states, st);
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail("hipStreamEndCapture failed: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&s->step_graph, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { s->step_graph = nullptr; return fail("hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
    s->graph_states = d_states;
    s->graph_flags = flags;
    return 0;
}

extern "C" int eaz_selfplay_run(eaz_search* s, const eaz_selfplay_cfg* cfg, uint64_t key, void* d_states, const float* d_gumbel,
                                void* d_out_states, float* d_out_root_value, float* d_out_root_std, float* d_out_value_pred,
                                float* d_out_ube_pred, float* d_out_qvar, int32_t* d_out_action, void* stream) {
    ARG_OK(s && cfg && d_states && cfg->selfplay_steps >= 0);
    if ((cfg->directed_exploration != 0) != (s->opts.exploration != 0))
        return fail("selfplay: directed_exploration (%d) must match the search's recurrent-fn `exploration` flag (%d), main.py:262",
                    cfg->directed_exploration, s->opts.exploration);
    HIP_OK(hipSetDevice(s->ctx->device));
    const int B = s->B, A = s->A;
    const size_t sb = static_cast<size_t>(s->env->dev.state_bytes) * B, fb = sizeof(float) * B;
    const Key k = key_from_u64(key);
    const int flags = (cfg->uniform_search_policy ? kFlagUniform : 0) | (cfg->sample_actions ? kFlagSampleVisits : 0) |
                      (cfg->sample_from_improved_policy ? kFlagSampleImproved : 0) | (d_gumbel ? kFlagExplicitGumbel : 0) |
                      (cfg->directed_exploration ? 16 : 0) | (static_cast<int>(cfg->exploration_beta != 0.0f) << 5);
    bool ds_fit = false;
    if (s->persist_ok && g_persist && cfg->selfplay_steps > 0) {
        // a hand-made state the kernel's packed state word cannot hold sends the call to the per-step kernels (search_persist.hpp ds_roots_fit_kernel)
        const int rc = ds_roots_fit(s, d_states, S(stream), &ds_fit);
        if (rc) return rc;
    }
    if (s->persist_ok && g_persist && (ds_fit || cfg->selfplay_steps == 0)) {
        // DeepSea + FC network: the WHOLE call -- T real steps of root inference, search, action choice, auto-reset step and output -- is ONE
        // launch of the persistent kernel (csrc/search_persist.hpp); nothing is captured, staged or copied
        if (cfg->selfplay_steps == 0) return 0;
        PersistSelfplay sp;
        std::memset(&sp, 0, sizeof sp);
        sp.on = 1; sp.T = cfg->selfplay_steps; sp.key = k;
        sp.flags = (cfg->uniform_search_policy ? kSpUniform : 0) | (cfg->sample_actions ? kSpSampleVisits : 0) |
                   (cfg->sample_from_improved_policy ? kSpSampleImproved : 0);
        sp.sbeta = cfg->directed_exploration ? cfg->exploration_beta : 0.0f;  // selfplay.py:92
        sp.gumbel_scale = s->opts.gumbel_scale;
        sp.states = static_cast<char*>(d_states); sp.gumbel = d_gumbel;
        sp.out_states = static_cast<char*>(d_out_states); sp.out_root_value = d_out_root_value; sp.out_root_std = d_out_root_std;
        sp.out_value_pred = d_out_value_pred; sp.out_ube_pred = d_out_ube_pred; sp.out_qvar = d_out_qvar; sp.out_action = d_out_action;
        s->last_sp = sp;
        ProfScope ps(s, EAZ_K_SEARCH, S(stream));
        return persist_launch(s, S(stream), &sp);
    }
    bool sq_fit = false;
    if (g_persist == 1 && sqp_eligible(s) && cfg->selfplay_steps > 0) {
        // a hand-made state beyond the compact form of the one-launch kernel (program words >= 16, IO values > 16, ...) sends the call to
        // the per-step path below, which keeps whole blobs
        const int rc = sq16_roots_fit(s, d_states, S(stream), &sq_fit);
        if (rc) return rc;
    }
    if (g_persist == 1 && sqp_eligible(s) && (sq_fit || cfg->selfplay_steps == 0)) {
        // Subleq-16 + FC network: likewise the whole call in ONE launch (csrc/search_subleq_persist.hpp)
        if (cfg->selfplay_steps == 0) return 0;
        PersistSelfplay sp;
        std::memset(&sp, 0, sizeof sp);
        sp.on = 1; sp.T = cfg->selfplay_steps; sp.key = k;
        sp.flags = (cfg->uniform_search_policy ? kSpUniform : 0) | (cfg->sample_actions ? kSpSampleVisits : 0) |
                   (cfg->sample_from_improved_policy ? kSpSampleImproved : 0);
        sp.sbeta = cfg->directed_exploration ? cfg->exploration_beta : 0.0f;  // selfplay.py:92
        sp.gumbel_scale = s->opts.gumbel_scale;
        sp.states = static_cast<char*>(d_states); sp.gumbel = d_gumbel;
        sp.out_states = static_cast<char*>(d_out_states); sp.out_root_value = d_out_root_value; sp.out_root_std = d_out_root_std;
        sp.out_value_pred = d_out_value_pred; sp.out_ube_pred = d_out_ube_pred; sp.out_qvar = d_out_qvar; sp.out_action = d_out_action;
        s->last_sp = sp;
        ProfScope ps(s, EAZ_K_SEARCH, S(stream));
        return sqp_launch(s, sp, S(stream));
    }
    // graph replay needs a capturable stream: work runs on the search's own stream, ordered after/before the caller's by events
    const bool use_graph = cfg->use_graph && !s->prof_on;
    hipStream_t user = S(stream), st = user;
    if (use_graph) {
        st = s->main;
        HIP_OK(hipEventRecord(s->ev_in, user));
        HIP_OK(hipStreamWaitEvent(st, s->ev_in, 0));
        if (s->graph_beta != cfg->exploration_beta) { s->graph_flags = -1; s->graph_beta = cfg->exploration_beta; }
    }
    for (int step = 0; step < cfg->selfplay_steps; ++step) {
        const Key kt = fold(k, kTagStep + static_cast<uint32_t>(step));  // key_seq[t], selfplay.py:147
        const long tb = static_cast<long>(step) * B;
        const StepOut slot{d_out_value_pred ? d_out_value_pred + tb : nullptr, d_out_ube_pred ? d_out_ube_pred + tb : nullptr,
                           d_out_root_value ? d_out_root_value + tb : nullptr, d_out_root_std ? d_out_root_std + tb : nullptr,
                           d_out_qvar ? d_out_qvar + tb * A : nullptr, d_out_action ? d_out_action + tb : nullptr,
                           d_out_states ? static_cast<char*>(d_out_states) + tb * s->env->dev.state_bytes : nullptr};
        set_step_keys_kernel<<<1, 1, 0, st>>>(s->d_step_keys, fold(kt, kTagGumbel), fold(kt, kTagEnv), fold(kt, kTagSampleVisits),
                                             fold(kt, kTagSampleImproved), s->d_step_out, slot);
        LAUNCH_OK();
        if (d_gumbel) HIP_OK(hipMemcpyAsync(s->d_gumbel_stage, d_gumbel + tb * A, fb * A, hipMemcpyDeviceToDevice, st));
        int rc;
        if (use_graph) {
            rc = selfplay_capture(s, cfg, flags, d_states, st);
            if (rc) return rc;
            HIP_OK(hipGraphLaunch(s->step_graph, st));
        } else {
            rc = selfplay_step_body(s, cfg, flags, d_states, st);
            if (rc) return rc;
        }
        // SelfplayOutput[t] (selfplay.py:136-143) was written by selfplay_emit_kernel, the last launch of the step body
    }
    if (use_graph) {
        HIP_OK(hipEventRecord(s->ev_out, st));
        HIP_OK(hipStreamWaitEvent(user, s->ev_out, 0));
    }
    return 0;
}

// root of evaluate / reanalyze: observation -> {novelty || heads} -> legal mask -> root_prep with a constant beta
static int driver_root(eaz_search* s, const void* d_states, float beta, const float* d_gumbel, Key gkey, hipStream_t st) {
    const int B = s->B, A = s->A;
    int rc = obs_launch(s->env, B, d_states, s->d_obs, st);
    if (rc) return rc;
    HeadList hl{3, {0, 1, 2, 0}};
    NetOut o{s->d_root_logits, nullptr, s->d_root_v, s->d_root_u};
    rc = infer_launch(s, s->d_obs, s->d_novelty, o, hl, st);
    if (rc) return rc;
    env_fields_kernel<<<cdiv(B, 256), 256, 0, st>>>(s->env->dev, static_cast<const char*>(d_states), B, s->d_term, nullptr, nullptr, nullptr,
                                                   s->d_legal, nullptr);
    LAUNCH_OK();
    set_step_keys_kernel<<<1, 1, 0, st>>>(s->d_step_keys, gkey, Key{0, 0}, Key{0, 0}, Key{0, 0}, nullptr, StepOut{});
    LAUNCH_OK();
    root_prep_kernel<<<cdiv(static_cast<long>(B) * A, 256), 256, 0, st>>>(B, A, s->d_root_logits, 0, beta, s->d_legal, d_gumbel, s->d_step_keys,
                                                                          s->opts.gumbel_scale, s->d_root_logits, s->d_root_beta,
                                                                          s->d_root_invalid, s->d_root_gumbel, s->d_root_u, s->d_novelty,
                                                                          s->net->dev.max_u, 1);
    LAUNCH_OK();
    return eaz_search_begin(s, s->d_root_logits, s->d_root_v, s->d_root_u, d_states, s->d_root_beta, s->d_root_invalid, s->d_root_gumbel, st);
}

extern "C" int eaz_evaluate_run(eaz_search* s, float exploitation_beta, int max_episode_length, void* d_states, float* d_sum_rewards,
                                int* h_iterations, void* stream) {
    ARG_OK(s && d_states && d_sum_rewards);
    if (s->opts.exploration) return fail("evaluate uses the exploitation recurrent fn (main.py:278-285): create the search with exploration = 0");
    if (s->opts.gumbel_scale != 0.0f) return fail("evaluate searches greedily (evaluate.py:44): create the search with gumbel_scale = 0");
    HIP_OK(hipSetDevice(s->ctx->device));
    hipStream_t st = S(stream);
    const int B = s->B;
    bool ds_fit = false;
    if (s->persist_ok && g_persist && max_episode_length >= 0 && max_episode_length < (1 << 24)) {
        const int rc = ds_roots_fit(s, d_states, st, &ds_fit);
        if (rc) return rc;
    }
    if (ds_fit) {
        // DeepSea: the whole while_loop (evaluate.py:13-57) as ONE launch of search_persist_kernel in its evaluate mode (search_persist.hpp)
        PersistSelfplay sp;
        std::memset(&sp, 0, sizeof sp);
        sp.on = 2; sp.T = max_episode_length + 1;  // cond_fn: counter <= max_episode_length
        sp.sbeta = exploitation_beta; sp.gumbel_scale = s->opts.gumbel_scale;
        sp.states = static_cast<char*>(d_states);
        sp.eval_sum = d_sum_rewards; sp.eval_iters = s->d_eval_wg; sp.eval_wg_iters = s->d_eval_wg + 1;
        HIP_OK(hipMemsetAsync(s->d_eval_wg, 0, sizeof(int) * (1 + cdiv(B, kPsTrees)), st));
        {
            ProfScope ps(s, EAZ_K_SEARCH, st);
            const int rc = persist_launch(s, st, &sp);
            if (rc) return rc;
        }
        {
            ProfScope ps(s, EAZ_K_MISC, st);
            eval_finalize_kernel<<<cdiv(B, 256), 256, 0, st>>>(static_cast<char*>(d_states), static_cast<long>(sizeof(eaz_deepsea_state)), B, kPsTrees,
                                                               s->d_eval_wg + 1, s->d_eval_wg);
            LAUNCH_OK();
        }
        if (h_iterations) {
            HIP_OK(hipMemcpyAsync(h_iterations, s->d_eval_wg, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
        }
        return 0;
    }
    if (g_persist == 1 && sqp_eligible(s) && max_episode_length >= 0 && max_episode_length < (1 << 24)) {
        // Subleq-16: the whole while_loop (evaluate.py:13-57) as ONE launch of the one-launch kernel in its evaluate mode -- no host round trip
        // between the first and the last step (the host driver below copies a counter back and synchronises after every real step)
        bool fit = false;
        int rc = sq16_roots_fit(s, d_states, st, &fit);
        if (rc) return rc;
        if (fit) {
            PersistSelfplay sp;
            std::memset(&sp, 0, sizeof sp);
            sp.on = 2; sp.T = max_episode_length + 1;  // cond_fn: counter <= max_episode_length
            sp.sbeta = exploitation_beta; sp.gumbel_scale = s->opts.gumbel_scale;
            sp.states = static_cast<char*>(d_states);
            sp.eval_sum = d_sum_rewards; sp.eval_iters = s->d_eval_wg; sp.eval_wg_iters = s->d_eval_wg + 1;
            HIP_OK(hipMemsetAsync(s->d_eval_wg, 0, sizeof(int) * (1 + cdiv(B, kQpTrees)), st));
            {
                ProfScope ps(s, EAZ_K_SEARCH, st);
                rc = sqp_launch(s, sp, st);
                if (rc) return rc;
            }
            {
                ProfScope ps(s, EAZ_K_MISC, st);
                eval_finalize_kernel<<<cdiv(B, 256), 256, 0, st>>>(static_cast<char*>(d_states), static_cast<long>(sizeof(eaz_subleq_state)), B, kQpTrees,
                                                                   s->d_eval_wg + 1, s->d_eval_wg);
                LAUNCH_OK();
            }
            if (h_iterations) {
                HIP_OK(hipMemcpyAsync(h_iterations, s->d_eval_wg, sizeof(int), hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
            }
            return 0;
        }
    }
    HIP_OK(hipMemsetAsync(d_sum_rewards, 0, sizeof(float) * B, st));
    int counter = 0;
    for (;;) {
        // cond_fn evaluate.py:18-20
        HIP_OK(hipMemsetAsync(s->d_alive, 0, sizeof(int), st));
        env_fields_kernel<<<cdiv(B, 256), 256, 0, st>>>(s->env->dev, static_cast<const char*>(d_states), B, s->d_term, nullptr, nullptr, nullptr, nullptr, nullptr);
        LAUNCH_OK();
        count_alive_kernel<<<cdiv(B, 256), 256, 0, st>>>(s->d_term, B, s->d_alive);
        LAUNCH_OK();
        int alive = 0;
        HIP_OK(hipMemcpyAsync(&alive, s->d_alive, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        if (alive == 0 || !(counter <= max_episode_length)) break;
        int rc = driver_root(s, d_states, exploitation_beta, nullptr, Key{0, 0}, st);  // :26-35 (gumbel_scale 0 -> zero noise)
        if (rc) return rc;
        rc = eaz_search_run(s, st);
        if (rc) return rc;
        rc = eaz_search_finish(s, s->d_action, s->d_aw, st);
        if (rc) return rc;
        rc = env_step_launch(s->env, B, d_states, s->d_action, Key{0, 0}, 0, d_states, nullptr, st, nullptr, s->d_reward_tmp);  // :47
        if (rc) return rc;
        accumulate_kernel<<<cdiv(B, 256), 256, 0, st>>>(d_sum_rewards, s->d_reward_tmp, B);  // :48-50
        LAUNCH_OK();
        counter += 1;
    }
    if (h_iterations) *h_iterations = counter;
    return 0;
}

extern "C" int eaz_reanalyze_run(eaz_search* s, const eaz_reanalyze_cfg* cfg, uint64_t key, const void* d_states, const void* d_next_states,
                                 const float* d_gumbel, float* d_value_target, float* d_ube_target, float* d_exploration_policy_target,
                                 float* d_exploitation_policy_target, int32_t* d_action, void* stream) {
    ARG_OK(s && cfg && d_states && d_next_states && d_value_target && d_ube_target && d_exploration_policy_target && d_exploitation_policy_target);
    if (s->opts.exploration) return fail("reanalyze uses the exploitation recurrent fn (main.py:270-277): create the search with exploration = 0");
    if (s->env->dev.kind == EAZ_ENV_CONNECT4) return fail("reanalyze: the reference's TD target squeezes a [B,1] reward (reanalyze.py:95), i.e. single-player envs only");
    HIP_OK(hipSetDevice(s->ctx->device));
    hipStream_t st = S(stream);
    const int B = s->B, A = s->A;
    int rc = driver_root(s, d_states, cfg->reanalyze_beta, d_gumbel, fold(key_from_u64(key), kTagGumbel), st);  // reanalyze.py:64-76
    if (rc) return rc;
    rc = eaz_search_run(s, st);                                                                               // :77-85
    if (rc) return rc;
    rc = eaz_search_finish(s, s->d_action, d_exploitation_policy_target, st);                                 // :129
    if (rc) return rc;
    rc = eaz_search_summary(s, s->d_sum_value, s->d_sum_std, s->d_sum_q, s->d_sum_qvar, s->d_sum_vc, nullptr, st);  // :86
    if (rc) return rc;
    // value of the next state (:90-92): only the value head is needed
    rc = obs_launch(s->env, B, d_next_states, s->d_obs, st);
    if (rc) return rc;
    {
        HeadList hl{1, {0, 0, 0, 0}};
        NetOut o{nullptr, nullptr, s->d_next_v, nullptr};
        rc = fc_launch(s->net, B, s->d_obs, o, hl, st, s->d_obs_perm);
        if (rc) return rc;
    }
    env_fields_kernel<<<cdiv(B, 256), 256, 0, st>>>(s->env->dev, static_cast<const char*>(d_next_states), B, s->d_term2, nullptr, s->d_reward_tmp,
                                                   nullptr, nullptr, nullptr);
    LAUNCH_OK();
#define CALL_RT(L, P)                                                                                                              \
    reanalyze_targets_kernel<L, P><<<tree_grid(s), 256, 0, st>>>(B, A, s->d_action, s->d_sum_q, s->d_sum_qvar, s->d_sum_vc, s->d_sum_value,  \
                                                                 s->d_sum_std, s->d_root_invalid, s->d_next_v, s->d_reward_tmp, s->d_term2,    \
                                                                 s->d_term, s->opts.discount, cfg->exploration_beta,                         \
                                                                 cfg->exploration_ube_target, cfg->exploration_policy_target_temperature,   \
                                                                 d_value_target, d_ube_target, d_exploration_policy_target)
    TREE_DISPATCH(s, CALL_RT);
#undef CALL_RT
    LAUNCH_OK();
    if (d_action) HIP_OK(hipMemcpyAsync(d_action, s->d_action, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int eaz_random_play(eaz_env* env, int B, int T, uint64_t key, void* d_states, void* d_traj, void* stream) {
    ARG_OK(env && d_states && B > 0 && T >= 0);
    hipStream_t st = S(stream);
    const Key k = key_from_u64(key);
    int rc = env_step_launch(env, B, nullptr, nullptr, fold(k, kTagInit), 2, d_states, nullptr, st);
    if (rc) return rc;
    int32_t* d_act = nullptr;
    HIP_OK(hipMallocAsync(reinterpret_cast<void**>(&d_act), sizeof(int32_t) * B, st));
    const size_t sb = static_cast<size_t>(env->dev.state_bytes) * B;
    auto steps = [&]() -> int {
        for (int step = 0; step < T; ++step) {
            const Key kt = fold(k, kTagStep + static_cast<uint32_t>(step));
            random_action_kernel<<<cdiv(B, 256), 256, 0, st>>>(B, env->dev.A, fold(kt, kTagRandomAction), d_act);
            LAUNCH_OK();
            if (d_traj) HIP_OK(hipMemcpyAsync(static_cast<char*>(d_traj) + sb * step, d_states, sb, hipMemcpyDeviceToDevice, st));
            const int rs = env_step_launch(env, B, d_states, d_act, fold(kt, kTagEnv), 1, d_states, nullptr, st);
            if (rs) return rs;
        }
        return 0;
    };
    rc = steps();
    const hipError_t fe = hipFreeAsync(d_act, st);  // also on the error path
    if (rc) return rc;
    HIP_OK(fe);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// instrumentation
// ------------------------------------------------------------------------------------------------------------------
#ifdef EAZ_FC_STAMPS
extern "C" __attribute__((visibility("default"))) int eaz_diag_tree_stamps(unsigned long long* h_out /* [1024*8] */) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_tree_stamps), sizeof(unsigned long long) * 1024 * 8));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_persist_stamps2(eaz_search* s, unsigned long long* h_out /* [256*32] */) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h_out, s->d_stamps + 256 * 32, sizeof(unsigned long long) * 256 * 32, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_persist_stamps(eaz_search* s, unsigned long long* h_out /* [256*32] */) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h_out, s->d_stamps, sizeof(unsigned long long) * 256 * 32, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_qp_stamps(eaz_search* s, unsigned long long* h_out /* [256*8*16] */) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h_out, s->d_stamps, sizeof(unsigned long long) * 256 * 8 * 16, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_qp_dhist(unsigned long long* h_out /* [72] */) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_qp_dhist), sizeof(unsigned long long) * 72));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_qp_fine(unsigned long long* h_out /* [2048*8] */) {
    ARG_OK(h_out != nullptr);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_qp_fine), sizeof(unsigned long long) * 2048 * 8));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_qp_its(uint32_t* h_out /* [64*64*16] */) {
    ARG_OK(h_out != nullptr);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_qp_its), sizeof(uint32_t) * 64 * 64 * 16));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_qp_fwd(unsigned long long* h_out /* [2048*8] */) {
    ARG_OK(h_out != nullptr);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_qp_fwd), sizeof(unsigned long long) * 2048 * 8));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_qp_rounds(unsigned long long* h_out /* [2048*64] */) {
    ARG_OK(h_out != nullptr);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_qp_rounds), sizeof(unsigned long long) * 2048 * 64));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_qp_ehist(unsigned long long* h_out /* [128] */, int reset) {
    ARG_OK(h_out != nullptr);
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_qp_ehist), sizeof(unsigned long long) * 128));
    if (reset) {
        static const unsigned long long zeros[128] = {};
        HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_qp_ehist), zeros, sizeof zeros));
    }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_sq_stamps(unsigned long long* h_out /* [16] */) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_sq_stamps), sizeof(unsigned long long) * 16));
    return 0;
}
extern "C" __attribute__((visibility("default"))) int eaz_diag_fc_stamps(unsigned long long* h_out /* [1024*8] */) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_fc_stamps), sizeof(unsigned long long) * 1024 * 8));
    return 0;
}
#endif

extern "C" int eaz_profile_enable(eaz_search* s, int on) {
    ARG_OK(s != nullptr);
    s->prof_on = on;
    s->net->d_kstep_counter = on ? s->net->d_kstep_storage : nullptr;
    return 0;
}

extern "C" int eaz_profile_net_ksteps(eaz_search* s, uint64_t* executed, int reset) {
    ARG_OK(s && executed);
    HIP_OK(hipDeviceSynchronize());
    unsigned long long v = 0;
    HIP_OK(hipMemcpy(&v, s->net->d_kstep_storage, sizeof v, hipMemcpyDeviceToHost));
    *executed = v;
    if (reset) HIP_OK(hipMemset(s->net->d_kstep_storage, 0, sizeof v));
    return 0;
}

extern "C" int eaz_profile_replay(eaz_search* s, int cls, int iters, double* ms_total, void* stream) {
    ARG_OK(s && ms_total && iters > 0 && (cls == EAZ_K_NET || cls == EAZ_K_GATHER || cls == EAZ_K_SEARCH));
    HIP_OK(hipSetDevice(s->ctx->device));
    hipStream_t st = S(stream);
    const bool fused = s->fuse_leaf != 0;
    if (cls == EAZ_K_GATHER && (fused || s->env->dev.kind != EAZ_ENV_DEEPSEA))
        return fail("eaz_profile_replay: the standalone leaf gather exists for unfused DeepSea searches only");
    HeadList hl{3, {0, 1, s->opts.exploration ? 3 : 2, 0}};
    NetOut o{s->opts.exploration ? nullptr : s->d_exploit, s->opts.exploration ? s->d_explore : nullptr, s->d_v, s->d_u};
    const bool sqp = g_persist == 1 && sqp_eligible(s);
    if (cls == EAZ_K_SEARCH && !(s->persist_ok && g_persist) && !(sqp && s->last_sp.on))
        return fail("eaz_profile_replay: this search does not run as one persistent launch (EAZ_TUNE_PERSIST, eligibility: eaz.h)");
    auto once = [&]() -> int {
        // the persistent search re-reads the root that the last eaz_search_begin left in the tree arrays (timing only: the statistics of
        // node 0 keep the previous launch's totals)
        if (cls == EAZ_K_SEARCH) {
            if (!s->last_sp.on) return persist_launch(s, st);
            // the last fused self-play call again, on a scratch copy of its states and without outputs
            PersistSelfplay sp = s->last_sp;
            sp.states = s->d_replay_states; sp.gumbel = nullptr;
            sp.out_states = nullptr; sp.out_root_value = sp.out_root_std = sp.out_value_pred = sp.out_ube_pred = sp.out_qvar = nullptr;
            sp.out_action = nullptr;
            if (sqp) return sqp_launch(s, sp, st);
            return persist_launch(s, st, &sp);
        }
        if (cls == EAZ_K_NET) return fused ? fc_onehot_launch(s->net, s->B, s->d_leaf_hot, s->d_novelty, o, hl, st) : fc_launch(s->net, s->B, s->d_obs, o, hl, st, s->d_obs_perm);
        leaf_gather_deepsea_kernel<<<cdiv(static_cast<long>(cdiv(s->B, kLgTrees)) * 64, 256), 256, 0, st>>>(s->env->dev, s->tree, s->d_obs, s->d_next_reward,
                                                                                          s->d_next_term, s->d_next_legal);
        LAUNCH_OK();
        return 0;
    };
    if (cls == EAZ_K_SEARCH && s->last_sp.on)
        HIP_OK(hipMemcpyAsync(s->d_replay_states, s->last_sp.states, static_cast<size_t>(s->B) * s->env->dev.state_bytes, hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < (cls == EAZ_K_SEARCH ? 1 : 3); ++i) { int rc = once(); if (rc) return rc; }
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
    HIP_OK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) { int rc = once(); if (rc) return rc; }
    HIP_OK(hipEventRecord(b, st));
    HIP_OK(hipEventSynchronize(b));
    float t = 0.0f;
    HIP_OK(hipEventElapsedTime(&t, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_total = t;
    return 0;
}

extern "C" int eaz_profile_read(eaz_search* s, double ms[EAZ_K_COUNT], int64_t launches[EAZ_K_COUNT], int reset) {
    ARG_OK(s && ms && launches);
    HIP_OK(hipDeviceSynchronize());
    for (int i = 0; i < EAZ_K_COUNT; ++i) { ms[i] = 0.0; launches[i] = 0; }
    for (auto& sp : s->spans) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) { ms[sp.cls] += t; launches[sp.cls] += 1; }
    }
    if (reset) {
        for (auto& sp : s->spans) { s->ev_pool.push_back(sp.a); s->ev_pool.push_back(sp.b); }
        s->spans.clear();
    }
    return 0;
}
