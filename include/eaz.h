/*
 * include/eaz.h -- C ABI of libeaz.so, the MI355X-native Epistemic-AlphaZero self-play engine.
 *
 * The reference (emcts/e-alphazero) is pure Python/JAX and has no FFI layer; its de-facto operator boundary for
 * the self-play hot path is three Python call shapes:
 *   (i)   pgx.Env.init / step / observe + State fields      src/selfplay.py:84,135,166 ; src/context.py:125-139
 *   (ii)  forward.apply(params, state, obs, is_training=False) -> (exploit_logits, explore_logits, v, u, nov)
 *                                                            src/selfplay.py:89-91 ; src/context.py:86-106
 *   (iii) emctx.epistemic_gumbel_muzero_policy(...) + search_tree.epistemic_summary(), with the recurrent fn of
 *         src/context.py:117-155                            src/selfplay.py:100-121
 * plus the fused driver selfplay() itself                    src/selfplay.py:78-149
 * Each entry point below names the call shape it replaces.  The Python facade in e-alphazero_amd/ binds these
 * with ctypes and re-exposes the reference's names; INTEGRATION.md shows the stub a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; eaz_last_error() returns a thread-local message.
 *   - "d_" pointers are DEVICE pointers owned by the caller (e.g. torch ROCm tensors); "h_" pointers are host.
 *   - all kernels are enqueued on the caller-supplied hipStream_t (passed as void*; NULL = default stream);
 *     calls are asynchronous unless stated otherwise.
 *   - one context per device; a context and the objects created from it are not thread-safe.
 *   - keys are 64-bit (k0 | k1 << 32) seeds of the build's Threefry-2x32 stream (see csrc/rng.hpp).
 *   - floats are IEEE binary32 everywhere ("dtype f32"); observations are uint8 (the reference's bool).
 */
#ifndef EAZ_H
#define EAZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAZ_API __attribute__((visibility("default")))

typedef struct eaz_ctx eaz_ctx;
typedef struct eaz_env eaz_env;
typedef struct eaz_net eaz_net;
typedef struct eaz_search eaz_search;

#define EAZ_ENV_DEEPSEA 0
#define EAZ_ENV_SUBLEQ 1
#define EAZ_ENV_CONNECT4 2

/* ---- compact env state blobs (AoS, one per environment instance) --------------------------------------------
 * The reference keeps the whole pgx.State (incl. the observation) per tree node (src/selfplay.py:104); the
 * observation is a pure function of these fields (deep_sea.py:66-70, subleq.py:697-705) and is regenerated
 * by the leaf-gather kernel instead of being stored. */
typedef struct {
    int32_t step_count; /* _step_count           deep_sea.py:22 */
    int32_t pos;        /* _horizontal_position  deep_sea.py:23 */
    float reward;       /* rewards[0]            deep_sea.py:16 */
    uint8_t terminated, truncated, pad[2];
} eaz_deepsea_state; /* 16 bytes */

#define EAZ_SUBLEQ_IO 8 /* MAXIMUM_INPUT_LENGTH = MAXIMUM_OUTPUT_LENGTH, subleq.py:15-17 */
typedef struct {
    int32_t step_count; /* _step_count subleq.py:554 */
    int32_t task;       /* _task :555 */
    float reward;       /* rewards[0] :549 */
    uint8_t terminated, truncated, solved, pad;
    uint16_t ex_in[EAZ_SUBLEQ_IO], ex_in_after[EAZ_SUBLEQ_IO], ex_out[EAZ_SUBLEQ_IO], ex_out_after[EAZ_SUBLEQ_IO]; /* :559-562 */
    uint8_t memory[256]; /* _memory_state :558 (only the first W bytes are meaningful; words are < W <= 256) */
} eaz_subleq_state; /* 336 bytes */

/* pgx connect_four [EXT] (BASELINE.json configs[3]): 6x7 board, row 0 at the top, -1 empty / 0 / 1 stone colour */
typedef struct {
    int32_t step_count;     /* _step_count */
    int32_t current_player; /* State.current_player: identity (0/1) of the player to move */
    float reward0;          /* rewards[0] */
    uint8_t terminated, truncated, color /* colour to move */, pad;
    float reward1;          /* rewards[1] */
    int32_t winner;         /* -1 or the winning colour */
    int8_t board[42];
    uint8_t pad2[14];
} eaz_connect4_state; /* 80 bytes */

/* ---- context ------------------------------------------------------------------------------------------------ */
EAZ_API const char* eaz_last_error(void);
EAZ_API int eaz_version(void);
/* first 12 hex digits of the SHA-256 over the .hip and .hpp files of csrc (sorted by name) and include/eaz.h this library was built from
   ("unknown" for a build outside csrc/Makefile): measurement files under profiles/ carry it, bench.py compares (no reference counterpart) */
EAZ_API const char* eaz_build_hash(void);
EAZ_API int eaz_ctx_create(int device, eaz_ctx** out);
EAZ_API int eaz_ctx_destroy(eaz_ctx* ctx);
/* child key = fold(key, tag) of the engine's Threefry-2x32 key derivation (csrc/rng.hpp); the counterpart of
 * jax.random.split / fold_in at src/selfplay.py:87,145-147 */
EAZ_API int eaz_key_fold(uint64_t key, uint32_t tag, uint64_t* out);

/* ---- (i) environments: pgx.Env surface -------------------------------------------------------------------- */
/* DeepSea(size_of_grid, action_map_key) deep_sea.py:37-52; the caller supplies the Bernoulli(0.5) map itself. */
EAZ_API int eaz_env_create_deepsea(eaz_ctx* ctx, int size_of_grid, const uint8_t* h_action_map /* [G*G] or NULL */, eaz_env** out);
/* Subleq(tasks, word_size, use_binary_encoding) subleq.py:599-621 */
EAZ_API int eaz_env_create_subleq(eaz_ctx* ctx, int word_size, const int32_t* h_tasks, int n_tasks, int use_binary_encoding, eaz_env** out);
/* Subleq(reward_fn=...) subleq.py:603: solved_or_not (:535-537, the default) or lowest_bytes (:540-542: solved / (1 + bytes_used),
 * bytes_used = the highest cursor + 3 any executed cycle of the three test runs reached, :311,:530).  Set before the first init / step. */
enum { EAZ_SUBLEQ_REWARD_SOLVED_OR_NOT = 0, EAZ_SUBLEQ_REWARD_LOWEST_BYTES = 1 };
EAZ_API int eaz_env_set_subleq_reward(eaz_env* env, int reward_fn);
/* Inspection (host only, no device): the test vectors the word_size-16 interpreter reads for `task`, test case c in 0..2 -- the
 * compile-time table of csrc/envs.hpp, i.e. get_test_cases (subleq.py:398-501) with the branch index clamped like lax.switch; 8 values
 * each, W = 16 beyond the valid ones.  tests/test_abi_and_host.py holds it against the oracle's generator for every task. */
EAZ_API int eaz_subleq16_test_vectors(int task, int c, uint16_t in[8], uint16_t expected_out[8]);
/* The same for any word_size in 16..256: the run-time generator of csrc/envs.hpp that the per-lane interpreter (word sizes other than 16)
 * evaluates on the device, compiled for the host as well. */
EAZ_API int eaz_subleq_test_vectors(int word_size, int task, int c, uint16_t in[8], uint16_t expected_out[8]);
/* pgx.make("connect_four") [EXT] main.py:190-194 */
EAZ_API int eaz_env_create_connect4(eaz_ctx* ctx, eaz_env** out);
EAZ_API int eaz_env_destroy(eaz_env* env);
/* num_actions, observation shape [rows, cols] and the byte size of one compact state */
EAZ_API int eaz_env_info(const eaz_env* env, int* kind, int* num_actions, int* obs_rows, int* obs_cols, int* state_bytes);
/* vmap(env.init)(keys)   selfplay.py:166 */
EAZ_API int eaz_env_init(eaz_env* env, int B, uint64_t key, void* d_states, void* stream);
/* vmap(env.step)(state, action, keys) context.py:127 ; autoreset != 0: vmap(auto_reset(env.step, env.init)) selfplay.py:26-75,135 */
EAZ_API int eaz_env_step(eaz_env* env, int B, const void* d_states_in, const int32_t* d_action, uint64_t key, int autoreset,
                 void* d_states_out, void* stream);
/* state.observation -> uint8 [B, rows*cols] */
EAZ_API int eaz_env_observe(eaz_env* env, int B, const void* d_states, uint8_t* d_obs, void* stream);
/* state.terminated / truncated / rewards [B, num_players] / _step_count / legal_action_mask [B,A] / current_player [B]
 * (any pointer may be NULL; num_players is 1 for DeepSea and Subleq, 2 for connect_four) */
EAZ_API int eaz_env_fields(eaz_env* env, int B, const void* d_states, uint8_t* d_terminated, uint8_t* d_truncated,
                   float* d_rewards, int32_t* d_step_count, uint8_t* d_legal_mask, int32_t* d_current_player, void* stream);
EAZ_API int eaz_env_num_players(const eaz_env* env);

/* ---- (ii) network: forward.apply(params, state, obs, is_training=False) ------------------------------------ */
/* EpistemicFullyConnectedAZNet fully_connected.py:14-39.  h_w[i] is [in,out] row-major, h_b[i] is [out], i in haiku
 * creation order: value 0-2, ube 3-5, exploitation 6-8, exploration 9-11 (fully_connected.py:50-81).
 * hash_off/hash_len select the flattened slice that is hashed (hash_io: rows word_size.. , fully_connected.py:85-87).
 * hidden = config.linear_layer_size (config.py:24): 256 runs the MFMA kernels and the one-launch self-play; 1..1024 otherwise: a plain kernel on
 * the per-simulation path (same results contract, not tuned). */
EAZ_API int eaz_net_create_fc(eaz_ctx* ctx, int in_dim, int hidden, int num_actions, const float* const* h_w, const float* const* h_b,
                      float max_ube, float max_reward_variance, int hash_off, int hash_len, int hash_bits, eaz_net** out);
/* EpistemicResidualAZNet(num_actions, num_channels=64, num_blocks) resnet.py:42-66 on [H,W,Cin] observations.  h_arrays: the
 * 42 + 12*num_blocks parameter/state arrays in haiku creation order (resnet.py:72-121): conv {w,b}; per block {bn, conv, bn,
 * conv}; final bn; policy {conv, bn, linear}; exploration {same}; value {conv, bn, linear, linear}; ube {same};
 * bn = {scale, offset, mean EMA, var EMA}, conv w = [kh,kw,in,out], linear w = [in,out]. */
EAZ_API int eaz_net_create_resnet(eaz_ctx* ctx, int H, int W, int Cin, int num_channels, int num_blocks, int num_actions,
                                  const float* const* h_arrays, int n_arrays, int hash_bits, eaz_net** out);
EAZ_API int eaz_net_destroy(eaz_net* net);
/* replace weights (after a learner update) -- same layout as eaz_net_create_fc */
EAZ_API int eaz_net_set_weights(eaz_net* net, const float* const* h_w, const float* const* h_b, void* stream);
/* Layer-1 INPUT ORDER of an FC network -- a switch of this library's numerics contract, no reference counterpart (XLA leaves the order of a
 * dot unspecified; forward.apply of src/network/fully_connected.py:45-81 is what is computed either way).  hk.Linear is evaluated as one
 * fmaf chain per output, acc = +0; acc = fmaf(x[k], w[k][j], acc); y = acc + b[j], over the inputs k ascending by default.  With an order
 * -- n_seg segments (seg_start[i], seg_len[i]) whose concatenation is a permutation of [0, in_dim) -- the chain visits the inputs in that
 * order instead; every kernel and the test oracle (eazo_net_set_input_order) follow, results differ from the default in the last bits only
 * (<= 1e-6 on the reference's networks).  Subleq uses {example_input, example_output, memory, input_after, output_after}: the example rows
 * are constants of an episode (src/envs/subleq.py:398-501,697-705), so the one-launch kernel starts every chain of a tree from one
 * precomputed partial sum and layer 1 runs over 144 instead of 240 inputs.  n_seg = 0 restores the default.  Not for one-hot (DeepSea)
 * observations.  Synchronises `stream`. */
EAZ_API int eaz_net_set_input_order(eaz_net* net, int n_seg, const int32_t* seg_start, const int32_t* seg_len, void* stream);
/* the same for a residual network: h_arrays as in eaz_net_create_resnet (same architecture).  The packed device buffer is rewritten in place
 * (same addresses and offsets), so searches and captured graphs that use this network see the new weights.  Synchronous. */
EAZ_API int eaz_net_set_resnet_weights(eaz_net* net, const float* const* h_arrays, int n_arrays, void* stream);
/* haiku state "fc_az_net/xxhash32/binary_set": uint8[2^(bits-3)] (hashes.py:40-43); host copy in/out */
EAZ_API int eaz_net_set_bitset(eaz_net* net, const uint8_t* h_bitset, void* stream);
EAZ_API int eaz_net_get_bitset(eaz_net* net, uint8_t* h_bitset, void* stream); /* synchronous */
/* outputs in the CODE's order (fully_connected.py:101): exploit[B,A], explore[B,A], v[B], u[B], novelty[B] (NULL = skip) */
EAZ_API int eaz_net_forward(eaz_net* net, int B, const uint8_t* d_obs, float* d_exploit, float* d_explore, float* d_value,
                    float* d_ube, float* d_novelty, void* stream);
/* hash_class = SimHash (config.py:27; hashes.py:100-139) instead of the default XXHash: h_random_matrix is the haiku state
 * "<net>/sim_hash/random_matrix", float [hash_len, hash_bits] row-major.  index = sum_j (x @ R)_j < 0 ? 2^j : 0, the product
 * accumulated as the hk.Linear chain (k ascending).  Applies to every later forward / lookup / update / search of this network;
 * set it before creating searches.  Synchronous.  (LCGHash, hashes.py:53-98, is self-declared broken and not provided.) */
EAZ_API int eaz_net_set_simhash(eaz_net* net, const float* h_random_matrix, void* stream);
/* BaseHash.__call__ / get_indices / update  hashes.py:23-50,162-229 (XXHash) or :124-139 (SimHash) on d_obs[:, hash_off:hash_off+hash_len] */
EAZ_API int eaz_hash_lookup(eaz_net* net, int B, const uint8_t* d_obs, uint8_t* d_seen, uint32_t* d_indices, void* stream);
EAZ_API int eaz_hash_update(eaz_net* net, int B, const uint8_t* d_obs, void* stream);

/* ---- (iii) search: emctx.epistemic_gumbel_muzero_policy --------------------------------------------------- */
typedef struct {
    int num_simulations;            /* selfplay.py:111 */
    int max_depth;                  /* emctx default None -> num_simulations (pass 0) */
    int max_num_considered_actions; /* emctx default 16 */
    float gumbel_scale;             /* emctx default 1.0 ; evaluate.py:44 passes 0.0 */
    /* qtransform = partial(epistemic_qtransform_completed_by_mix_value, rescale_values=...) selfplay.py:114-116 */
    int rescale_values;
    int use_mixed_value; /* 1 */
    float value_scale;   /* 0.1 */
    float maxvisit_init; /* 50 */
    /* recurrent fn factory arguments, context.py:109-115 */
    float discount;
    int two_players_game;
    int exploration; /* use exploration logits inside the tree (context.py:132) */
    /* The two choices of emctx's epistemic extension that cannot be checked offline (emctx is not vendored and not importable here,
     * SURVEY.md A.5 / A.6 "(?)"): both are switches, so adopting vectors from a real emctx run is a flag flip, and the oracle
     * (eazo_search_set_epistemic_variant) and tests/mctx_numpy.py follow the same switches.  With beta == 0 (the reference default,
     * config.py:38) selection does not depend on either; the variance outputs depend on the second one only.
     *   epistemic_bonus_after_rescale  0 (default): completed Q + beta * sigma, THEN the min-max rescale;  1: rescale the completed Q, then
     *                                  add beta * sigma (both before the (maxvisit_init + max N) * value_scale factor)
     *   epistemic_variance_backup      0 (default): node variance = running mean of r_var + gamma^2 * U over the visits (the twin of the
     *                                  value backup);  1: variance of that running mean, (var * n^2 + U) / (n + 1)^2 */
    int epistemic_bonus_after_rescale;
    int epistemic_variance_backup;
} eaz_search_opts;

EAZ_API int eaz_search_create(eaz_ctx* ctx, eaz_env* env, eaz_net* net, int B, const eaz_search_opts* opts, eaz_search** out);
EAZ_API int eaz_search_destroy(eaz_search* s);

/* EpistemicRootFnOutput(prior_logits, value, value_epistemic_variance, embedding, beta) selfplay.py:100-106,
 * invalid_actions (uint8 [B,A] or NULL), explicit root Gumbel noise (already multiplied by gumbel_scale). */
EAZ_API int eaz_search_begin(eaz_search* s, const float* d_prior_logits, const float* d_value, const float* d_value_var,
                     const void* d_root_states, const float* d_beta, const uint8_t* d_invalid, const float* d_gumbel,
                     void* stream);
/* one simulation = select -> leaf gather + env.step -> forward -> expand -> backward (fused native recurrent fn) */
EAZ_API int eaz_search_simulate(eaz_search* s, int sim, void* stream);
/* all simulations (sim = 0..S-1) */
EAZ_API int eaz_search_run(eaz_search* s, void* stream);
/* PolicyOutput.action [B] int32, action_weights [B,A] */
EAZ_API int eaz_search_finish(eaz_search* s, int32_t* d_action, float* d_action_weights, void* stream);
/* search_tree.epistemic_summary(): value[B], value_epistemic_std[B], qvalues[B,A], qvalues_epistemic_variance[B,A],
 * visit_counts[B,A] (f32), visit_probs[B,A]; any pointer may be NULL.  selfplay.py:119-121,125,142 ; reanalyze.py:86-116 */
EAZ_API int eaz_search_summary(eaz_search* s, float* d_value, float* d_value_std, float* d_qvalues, float* d_qvalues_var,
                       float* d_visit_counts, float* d_visit_probs, void* stream);

/* generic-recurrent-fn path: the caller evaluates recurrent_fn itself between the two halves.
 * select: writes parent_index[B], action[B], next_node_index[B]. */
EAZ_API int eaz_search_select(eaz_search* s, int sim, int32_t* d_parent, int32_t* d_action, int32_t* d_next, void* stream);
/* expand + backward with EpistemicRecurrentFnOutput(reward, reward_epistemic_variance, discount, prior_logits, value,
 * value_epistemic_variance) context.py:146-153 (prior_logits already masked by the caller's recurrent fn). */
EAZ_API int eaz_search_expand(eaz_search* s, int sim, const float* d_reward, const float* d_reward_var, const float* d_discount,
                      const float* d_prior_logits, const float* d_value, const float* d_value_var, void* stream);

/* tree field ids for eaz_search_tree_view (mctx Tree / emctx EpistemicTree field names) */
enum {
    EAZ_F_NODE_VISITS = 0, EAZ_F_RAW_VALUES = 1, EAZ_F_NODE_VALUES = 2, EAZ_F_PARENTS = 3, EAZ_F_ACTION_FROM_PARENT = 4,
    EAZ_F_CHILDREN_INDEX = 5, EAZ_F_CHILDREN_PRIOR_LOGITS = 6, EAZ_F_CHILDREN_VALUES = 7, EAZ_F_CHILDREN_VISITS = 8,
    EAZ_F_CHILDREN_REWARDS = 9, EAZ_F_CHILDREN_DISCOUNTS = 10, EAZ_F_RAW_VALUES_VAR = 11, EAZ_F_NODE_VALUES_VAR = 12,
    EAZ_F_CHILDREN_VALUES_VAR = 13, EAZ_F_CHILDREN_REWARDS_VAR = 14, EAZ_F_EMBEDDINGS = 15, EAZ_F_ROOT_GUMBEL = 16,
    EAZ_F_BETA = 17, EAZ_F_COUNT = 18
};
/* device pointer + shape ([B,N], [B,N,A], [B,N,state_bytes], [B,A] or [B]; unused dims = 0) + element size.  The arrays hold the tree of the
 * last eaz_search_begin / _simulate / _run / _expand sequence (every field of every node slot, mctx's initial values where a slot was never
 * expanded).  After eaz_selfplay_run / eaz_evaluate_run they are SCRATCH (the one-launch kernels never touch them; the per-step driver leaves
 * the tree of its last real step). */
EAZ_API int eaz_search_tree_view(eaz_search* s, int field_id, void** d_ptr, int64_t shape[3], int* elem_bytes);

/* TEST HOOK (no reference counterpart): fills scratch of the search with `byte_value` -- the content a freed allocation may leave behind.
 * what = 0: the select scratch (the recorded root-to-leaf paths, their lengths, the selected (parent, action, next) triples).  Between
 *   eaz_search_begin and eaz_search_run it reproduces the precondition of round 3's 257-second launch (a lane group beyond a ragged batch read
 *   another workgroup's path length before it was written): the search must finish in its normal time with unchanged results.
 * what = 1: the eight [B,N,A] child arrays of the tree.  In front of eaz_selfplay_run with lazy child rows (EAZ_TUNE_LAZY_ROWS) the call's
 *   results must not change: a search that never clears those arrays may not read an entry it has not written. */
EAZ_API int eaz_search_debug_poison(eaz_search* s, int what, int byte_value, void* stream);
/* Launch structure the library will use for this search under the current tuning (EAZ_TUNE_PERSIST): *one_launch = 1 when eaz_selfplay_run
 * (the whole call) and eaz_search_run (all simulations) each run as ONE kernel launch on LDS-resident trees -- DeepSea + FC network with two
 * actions (csrc/search_persist.hpp) or Subleq word_size 16, binary observations, FC network, XXHash novelty over the IO rows
 * (csrc/search_subleq_persist.hpp), trees small enough for the 160 KiB of LDS -- and 0 when the per-simulation kernels (or, Subleq-16 with
 * larger trees, csrc/search_subleq.hpp) serve it.  *lds_bytes = dynamic LDS of that launch (0 if none).  No reference counterpart: the
 * reference's search is one XLA program whatever the sizes; tests and benchmarks use this to know which path they measured. */
EAZ_API int eaz_search_launch_info(const eaz_search* s, int* one_launch, int* lds_bytes);

/* ---- fused driver: selfplay()  src/selfplay.py:78-149 ------------------------------------------------------ */
typedef struct {
    int selfplay_steps;               /* T, config.py:37 */
    int directed_exploration;         /* config.py:38 */
    float exploration_beta;           /* config.py:69 */
    int uniform_search_policy;        /* config.py:42 */
    int sample_actions;               /* config.py:39 */
    int sample_from_improved_policy;  /* config.py:40 */
    int use_graph;                    /* capture each real step into a hipGraph and replay it */
} eaz_selfplay_cfg;

/* d_states [B] compact states: in = last_states, out = new last_states.  d_gumbel: optional explicit unscaled root
 * noise [T,B,A] (NULL = generated from key).  Outputs are SelfplayOutput fields (selfplay.py:17-23), [T,B(,A)]:
 * state (compact blobs), root_value, root_epistemic_std, value_prediction, ube_prediction,
 * q_values_epistemic_variance; d_out_action [T,B] is an extra (the chosen actions).  Any output may be NULL. */
EAZ_API int eaz_selfplay_run(eaz_search* s, const eaz_selfplay_cfg* cfg, uint64_t key, void* d_states, const float* d_gumbel,
                     void* d_out_states, float* d_out_root_value, float* d_out_root_std, float* d_out_value_pred,
                     float* d_out_ube_pred, float* d_out_qvar, int32_t* d_out_action, void* stream);

/* uniformrandomplay() src/selfplay.py:152-170: init + T uniformly random auto-reset steps; d_states [B] out (last states),
 * d_traj optional [T,B] states before each step. */
EAZ_API int eaz_random_play(eaz_env* env, int B, int T, uint64_t key, void* d_states, void* d_traj, void* stream);

/* ---- "next" rows of the hot-path table: evaluate and reanalyze drivers (same kernels, different roots) --------------- */
/* evaluate() src/evaluate.py:13-57: greedy search episodes (the search must have been created with gumbel_scale = 0 and
 * exploration = 0), beta = exploitation_beta for every env (:34), plain env.step without auto-reset (:47), until every env
 * is terminated or counter > max_episode_length (:18-20).  d_states: vmap(env.init) states, advanced in place.
 * d_sum_rewards [B]: per-episode return (the reference returns its mean, :57).  *h_iterations: loop trips executed.
 * Synchronous (the loop condition is read back every iteration). */
EAZ_API int eaz_evaluate_run(eaz_search* s, float exploitation_beta, int max_episode_length, void* d_states, float* d_sum_rewards,
                             int* h_iterations, void* stream);

typedef struct {
    float reanalyze_beta;                        /* config.py:75, beta of every root (reanalyze.py:75) */
    float exploration_beta;                      /* config.py:69, used in the exploration policy target (reanalyze.py:112-118) */
    int exploration_ube_target;                  /* config.py:33 */
    float exploration_policy_target_temperature; /* config.py:66 */
} eaz_reanalyze_cfg;

/* reanalyze() src/reanalyze.py:52-131 for B (state, next_state) pairs: search from `states` with the exploitation logits
 * (search created with exploration = 0 and reanalyze_simulations_per_step), then
 *   value_target = max(Q_tree[action], r' + gamma * v(next) * !term') * !term      (:88-109)
 *   ube_target   = (max_a | [action]) qvalues_epistemic_variance * !term           (:103-110)
 *   exploration_policy_target = softmax(mask(complete(Q + beta*sigma)) * temperature)   (:112-122)
 *   exploitation_policy_target = action_weights                                     (:129)
 * d_gumbel: optional explicit unscaled root noise [B,A] (NULL = generated from key).  d_action optional [B]. */
EAZ_API int eaz_reanalyze_run(eaz_search* s, const eaz_reanalyze_cfg* cfg, uint64_t key, const void* d_states, const void* d_next_states,
                              const float* d_gumbel, float* d_value_target, float* d_ube_target, float* d_exploration_policy_target,
                              float* d_exploitation_policy_target, int32_t* d_action, void* stream);

/* ---- instrumentation -------------------------------------------------------------------------------------- */
/* per-kernel-class device time accumulated with hipEvents on the launch stream while profiling is enabled */
/* EAZ_K_SEARCH: the persistent search kernel (all simulations of a real step in one launch: tree + env step + network forward) */
enum { EAZ_K_TREE = 0, EAZ_K_GATHER = 1, EAZ_K_HASH = 2, EAZ_K_NET = 3, EAZ_K_MISC = 4, EAZ_K_SEARCH = 5, EAZ_K_COUNT = 6 };
EAZ_API int eaz_profile_enable(eaz_search* s, int on);
/* synchronises the stream; returns total ms and launch count per class since the last reset */
EAZ_API int eaz_profile_read(eaz_search* s, double ms[EAZ_K_COUNT], int64_t launches[EAZ_K_COUNT], int reset);
/* layer-1 k-steps (16 rows x 256 columns x 4 k each) the FC kernel actually executed while profiling was enabled: all-zero
 * observation k-steps are skipped exactly, so executed FLOPs <= dense FLOPs */
EAZ_API int eaz_profile_net_ksteps(eaz_search* s, uint64_t* executed, int reset);

/* Kernel duration without the launch gap an event pair around a single launch includes: the network forward (EAZ_K_NET) or
 * the standalone DeepSea leaf gather (EAZ_K_GATHER, unfused searches) of the LAST simulation, or the persistent search kernel
 * (EAZ_K_SEARCH: all simulations from the root of the last eaz_search_begin), is launched `iters` times back to back on `stream`
 * inside ONE event pair; *ms_total / iters is what rocprofv3 reports as the kernel's average duration. */
EAZ_API int eaz_profile_replay(eaz_search* s, int kernel_class, int iters, double* ms_total, void* stream);

/* Kernel-selection knobs for A/B measurements and for the parity tests (every setting computes identical results).  They are
 * read when a kernel is launched -- a real step that eaz_selfplay_run has already captured into its graph keeps the values it was
 * captured with -- so set them before creating the search they are meant for.
 *   EAZ_TUNE_FC_TILE    0 = automatic (the pipelined 16-row kernel for narrow heads on observations of <= 1024 inputs, otherwise 16-row
 *                       tiles while the launch is fully resident, else 64), 1 = pipelined where eligible, 16 or 64 = pinned
 *   EAZ_TUNE_FUSED_LEAF 1 = DeepSea searches created from now on step the environment inside the tree kernel and feed the
 *                           network the one-hot index (no leaf-gather launch); 0 = separate leaf-gather + dense forward */
#define EAZ_TUNE_FC_TILE 0
#define EAZ_TUNE_FUSED_LEAF 1
/*   EAZ_TUNE_TREE_BLOCK  threads per workgroup of tree_step_kernel (64, 128, 256)
 *   EAZ_TUNE_TREE_SPREAD lanes per tree = lanes-per-tree x spread (1, 2, 4, 8): fewer trees per wavefront, more wavefronts;
 *                        0 = automatic (spread until the launch has one wavefront per CU) */
#define EAZ_TUNE_TREE_BLOCK 2
#define EAZ_TUNE_TREE_SPREAD 3
/*   EAZ_TUNE_PERSIST     1 (default) = eaz_search_run runs all simulations in ONE launch, 16 trees per workgroup, where a kernel for it
 *                        exists, and eaz_selfplay_run the WHOLE call (T real steps) in one launch: DeepSea + FC network with two actions
 *                        whose trees fit into LDS (csrc/search_persist.hpp), and Subleq word_size 16 with binary observations + FC network
 *                        with the XXHash novelty over the IO rows (csrc/search_subleq_persist.hpp, trees of <= 69 nodes in LDS, for
 *                        both entry points; larger trees: csrc/search_subleq.hpp, eaz_search_run only); 2 = one launch per SEARCH only (Subleq-16: the per-real-step kernels
 *                        around csrc/search_subleq.hpp's kernel; DeepSea: as 1); 0 = tree kernel(s) + one forward per simulation.
 *                        eaz_search_simulate always uses the per-simulation kernels. */
#define EAZ_TUNE_PERSIST 4
/*   EAZ_TUNE_SHARDS      sub-batch pipelining of the per-simulation path, read by eaz_search_create: 2 / 4 = the batch is searched as that many
 *                        sub-batches on as many streams (one sub-batch's env-step launch overlaps another's forward), 1 = off,
 *                        0 (default) = 4 for Subleq batches of >= 2048 environments, 2 from 1024, else 1 (not used by the one-launch searches) */
#define EAZ_TUNE_SHARDS 5
/*   EAZ_TUNE_SUBLEQ_WAVE Subleq word sizes other than 16 on the per-simulation path: 1 (default) = one wavefront per environment
 *                        (subleq_wave_kernel, program in the lanes of one VGPR), 0 = one lane per environment (subleq_step_kernel, program in LDS);
 *                        initial value from the environment variable EAZ_SUBLEQ_WAVE */
#define EAZ_TUNE_SUBLEQ_WAVE 6
/*   EAZ_TUNE_LAZY_ROWS   the per-step driver of eaz_selfplay_run (not the one-launch kernels): 1 (default) = searches over >= 64 actions do not
 *                        clear the eight [B,N,A] tree arrays per real step; a node's children_visits row is zeroed by the kernel that creates the
 *                        node and an edge with no visits reads as mctx's initial values (csrc/tree.hpp TreeDev::lazy_rows: same results);
 *                        2 = for every action count, 0 = off.  Initial value from the environment variable EAZ_LAZY_ROWS */
#define EAZ_TUNE_LAZY_ROWS 7
EAZ_API int eaz_tune_set(int knob, int value);

#ifdef __cplusplus
}
#endif
#endif /* EAZ_H */
