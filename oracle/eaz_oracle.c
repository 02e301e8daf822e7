/*
 * oracle/eaz_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the reference's self-play hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (libeaz.so + e-alphazero_amd/) never does.
 *
 * PARITY STATUS.  The reference is Python/JAX and cannot run here: jax, pgx, emctx and
 * dm-haiku are absent and unpinned (Pipfile:7-15).  Consequently
 *   - the parts restated from files under /root/reference (DeepSea, Subleq, FC net, XXHash variant, recurrent
 *     fn, auto-reset, selfplay driver) are pinned by the known-answer material those files hold (encoder
 *     docstrings, Subleq test tables, DeepSea invariants; tests/test_oracle_*.py);
 *   - the parts that live in the un-vendored third-party packages emctx (fork of google-deepmind/mctx,
 *     Pipfile:7, moving main.zip) and pgx (Pipfile:8, "*") are restated from their published algorithm
 *     (Gumbel MuZero: "Policy improvement by planning with Gumbel"; E-MCTS: arXiv 2210.13455) and anchored on
 *     the reference's call sites.  For those:  **parity unpinned**  (no golden vectors from the real emctx exist).
 *
 * Every function cites the reference file:line it follows.  Float operations are written out in a fixed order
 * (see o_math.h) so the HIP path can be compared bit-for-bit on integers and to 1e-5 on floats.
 */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "o_math.h"
#include "o_rng.h"

#define O_MAX_A 256
#define O_SUBLEQ_MAXW 256
#define O_IOLEN 8          /* MAXIMUM_INPUT_LENGTH == MAXIMUM_OUTPUT_LENGTH, src/envs/subleq.py:15-17 */
#define O_MAX_CYCLES 200   /* MAX_CYCLE_COUNT, src/envs/subleq.py:19 */
#define O_F32_MIN (-3.40282347e+38f) /* jnp.finfo(float32).min, src/context.py:137 */

#define EXPORT __attribute__((visibility("default")))

/* ===================================================================================================== *
 *  Environments                                                                                         *
 * ===================================================================================================== */

enum { O_ENV_DEEPSEA = 0, O_ENV_SUBLEQ = 1, O_ENV_CONNECT4 = 2 };

typedef struct OEnv {
    int kind;
    int num_actions;
    int obs_rows, obs_cols; /* obs_dim = rows*cols */
    int state_bytes;
    /* DeepSea (src/envs/deep_sea.py:37-52) */
    int G;
    uint8_t* action_map; /* [G*G] bool */
    /* Subleq (src/envs/subleq.py:599-621) */
    int W;
    int n_tasks;
    int tasks[16];
    int binary;
    int enc_width;
    int reward_fn; /* 0 = solved_or_not (subleq.py:535-537), 1 = lowest_bytes (:540-542) */
} OEnv;

/* compact state blobs; the observation is a pure function of these (deep_sea.py:66-70, subleq.py:697-705) */
typedef struct {
    int32_t step_count; /* _step_count */
    int32_t pos;        /* _horizontal_position */
    float reward;       /* rewards[0] */
    uint8_t terminated, truncated, pad[2];
} ODeepSeaState;

typedef struct {
    int32_t step_count;
    int32_t task;
    float reward;
    uint8_t terminated, truncated, solved, pad;
    int32_t memory[O_SUBLEQ_MAXW];
    int32_t ex_in[O_IOLEN], ex_in_after[O_IOLEN], ex_out[O_IOLEN], ex_out_after[O_IOLEN];
} OSubleqState;

/* ---- DeepSea ---------------------------------------------------------------------------------------- */

/* src/envs/deep_sea.py:54-57 (_init) */
static void o_deepsea_init(const OEnv* e, ODeepSeaState* s) {
    (void)e;
    memset(s, 0, sizeof(*s));
}

/* src/envs/deep_sea.py:59-81 (_step); step_count has already been incremented by the pgx wrapper */
static void o_deepsea_step(const OEnv* e, ODeepSeaState* s, int action) {
    const int G = e->G;
    /* jnp indexing clamps out-of-range indices; step_count-1 and pos are always in range for states the environment produced */
    int r = s->step_count - 1;
    if (r < 0) r = 0;
    if (r > G - 1) r = G - 1;
    int pc = s->pos < 0 ? 0 : (s->pos > G - 1 ? G - 1 : s->pos); /* a hand-made blob's column: clamped like the row */
    int flip = e->action_map[r * G + pc] != 0;
    int shift = ((action == 0) ^ flip) ? -1 : 1;
    int pos = s->pos + shift;
    if (pos < 0) pos = 0;
    if (pos > G - 1) pos = G - 1;
    int terminated = s->step_count >= G - 1;
    s->pos = pos;
    s->terminated = (uint8_t)terminated;
    s->reward = (terminated && pos == G - 1) ? 1.0f : 0.0f;
}

/* src/envs/deep_sea.py:66-70,83-85: one-hot at [min(step_count, G-1), pos] */
static void o_deepsea_observe(const OEnv* e, const ODeepSeaState* s, uint8_t* obs) {
    const int G = e->G;
    memset(obs, 0, (size_t)G * G);
    int r = s->step_count < G - 1 ? s->step_count : G - 1;
    if (r >= 0 && s->pos >= 0 && s->pos < G) obs[r * G + s->pos] = 1; /* (a hand-made blob off the grid addresses no cell) */
}

/* ---- Subleq ----------------------------------------------------------------------------------------- */

static int o_pymod(int x, int m) { int r = x % m; return r < 0 ? r + m : r; }

/* src/envs/subleq.py:101-107 (pad) + :402-404 (prepare: modulo then pad) */
static void o_prepare(const int* v, int n, int W, int32_t* out) {
    for (int i = 0; i < O_IOLEN; ++i) out[i] = W;
    for (int i = 0; i < n; ++i) out[i] = o_pymod(v[i], W);
}

/* src/envs/subleq.py:398-501 (get_test_cases).  lax.switch clamps the branch index (:459-501). */
static void o_subleq_test_cases(int task, int W, int32_t tin[3][O_IOLEN], int32_t tout[3][O_IOLEN]) {
    int branch = task - 1;
    if (branch < 0) branch = 0;
    if (branch > 5) branch = 5;
    int in[3][8], out[3][8], nin[3], nout[3];
    memset(in, 0, sizeof in);
    memset(out, 0, sizeof out);
    const int np1[8] = {5, 4, 4, 5, 1, 2, 3, 1}, np2[8] = {1, 1, 6, 2, 4, 4, 5, 3};
    const int ng2[5] = {0, -1, 2, -3, 4};
    switch (branch) {
        case 0: /* NEGATION_POSITIVE :406-411 */
            for (int i = 0; i < 7; ++i) in[0][i] = i + 1;
            nin[0] = 7;
            memcpy(in[1], np1, sizeof np1); nin[1] = 8;
            memcpy(in[2], np2, sizeof np2); nin[2] = 8;
            for (int c = 0; c < 3; ++c) { nout[c] = nin[c]; for (int i = 0; i < nin[c]; ++i) out[c][i] = -in[c][i]; }
            break;
        case 1: /* NEGATION :413-418 */
        case 2: /* IDENTITY :420-421 */
            for (int i = 0; i < 8; ++i) in[0][i] = -4 + i;
            nin[0] = 8;
            for (int i = 0; i < 7; ++i) in[1][i] = i + 1;
            nin[1] = 7;
            memcpy(in[2], ng2, sizeof ng2); nin[2] = 5;
            for (int c = 0; c < 3; ++c) {
                nout[c] = nin[c];
                for (int i = 0; i < nin[c]; ++i) out[c][i] = branch == 1 ? -in[c][i] : in[c][i];
            }
            break;
        case 3:   /* SUBTRACTION :423-432 */
        case 4: { /* ADDITION :434-443 */
            const int a0[6] = {1, 1, 5, 4, 0, -3}, a1[6] = {2, 3, 0, 0, -1, -2}, a2[8] = {1, 2, 3, 4, 4, 3, 2, 1};
            memcpy(in[0], a0, sizeof a0); nin[0] = 6;
            memcpy(in[1], a1, sizeof a1); nin[1] = 6;
            memcpy(in[2], a2, sizeof a2); nin[2] = 8;
            if (branch == 3) {
                const int o0[3] = {0, 1, 3}, o1[3] = {-1, 0, 1}, o2[4] = {-1, -1, 1, 1};
                memcpy(out[0], o0, sizeof o0); nout[0] = 3;
                memcpy(out[1], o1, sizeof o1); nout[1] = 3;
                memcpy(out[2], o2, sizeof o2); nout[2] = 4;
            } else {
                const int o0[3] = {2, 9, -3}, o1[3] = {5, 0, -3}, o2[4] = {3, 7, 7, 3};
                memcpy(out[0], o0, sizeof o0); nout[0] = 3;
                memcpy(out[1], o1, sizeof o1); nout[1] = 3;
                memcpy(out[2], o2, sizeof o2); nout[2] = 4;
            }
            break;
        }
        default: { /* MULTIPLICATION (and every clamped task id) :445-454 */
            const int m0[6] = {1, 2, 2, 3, -7, 4}, m1[6] = {-1, -5, 5, 2, 0, 1}, m2[6] = {0, 0, 1, 1, -3, 3};
            const int o0[3] = {2, 6, -28}, o1[3] = {5, 10, 0}, o2[3] = {0, 1, -9};
            memcpy(in[0], m0, sizeof m0); memcpy(in[1], m1, sizeof m1); memcpy(in[2], m2, sizeof m2);
            memcpy(out[0], o0, sizeof o0); memcpy(out[1], o1, sizeof o1); memcpy(out[2], o2, sizeof o2);
            for (int c = 0; c < 3; ++c) { nin[c] = 6; nout[c] = 3; }
            break;
        }
    }
    for (int c = 0; c < 3; ++c) {
        o_prepare(in[c], nin[c], W, tin[c]);
        o_prepare(out[c], nout[c], W, tout[c]);
    }
}

typedef struct {
    int32_t in_after[O_IOLEN], out_after[O_IOLEN];
    int bytes_used, cycles, correct;
} OSimResult;

/* src/envs/subleq.py:156-395 (simulate): one program run against one (input, expected output) pair */
static void o_subleq_simulate(int W, const int32_t* mem0, const int32_t* tin, const int32_t* tout, OSimResult* res) {
    const int AMAX = W - 4, AIN = W - 3, AOUT = W - 2; /* :162-165; W-1 = HALT */
    int32_t mem[O_SUBLEQ_MAXW], in[O_IOLEN], out[O_IOLEN];
    memcpy(mem, mem0, sizeof(int32_t) * (size_t)W);
    memcpy(in, tin, sizeof in);
    for (int i = 0; i < O_IOLEN; ++i) out[i] = W; /* :382 */
    int out_cursor = 0, cur = 0, bytes_used = 0, cycles = 0, halt = 0, error = 0;
    while (!error && !halt && cycles < O_MAX_CYCLES) { /* :297-299 */
        if (cur + 2 >= W) { /* :364-370 */
            cycles += 1;
            error = 1;
            continue;
        }
        cycles += 1;
        if (cur + 3 > bytes_used) bytes_used = cur + 3; /* :311 */
        int a = mem[cur], b = mem[cur + 1], c = mem[cur + 2];
        /* lax.switch clamps the address (:232,:295) */
        int aa = a < 0 ? 0 : (a > W - 1 ? W - 1 : a);
        int bb = b < 0 ? 0 : (b > W - 1 ? W - 1 : b);
        /* read_memory :187-232 -- both reads see the same (pre-advance) input state */
        int ra = 0, rb = 0, acc_a = 0, acc_b = 0, err_a = 0, err_b = 0;
        if (aa <= AMAX) ra = mem[aa];
        else if (aa == AIN) { if (in[0] >= W) err_a = 1; else { ra = in[0]; acc_a = 1; } }
        if (bb <= AMAX) rb = mem[bb];
        else if (bb == AIN) { if (in[0] >= W) err_b = 1; else { rb = in[0]; acc_b = 1; } }
        int value = o_pymod(ra - rb, W); /* :322 */
        /* write_memory :243-295 */
        int werr = 0, modified = 0;
        if (aa <= AMAX) mem[aa] = value;
        else if (aa == AOUT) {
            if (out_cursor >= O_IOLEN) werr = 1;
            else { out[out_cursor] = value; out_cursor += 1; modified = 1; }
        }
        int jump = (value == 0) || (2 * value >= W); /* :329  value >= word_size / 2 */
        cur = jump ? c : cur + 3;                    /* :330 */
        if (acc_a || acc_b) {                        /* :333-338 roll(-1), last = W */
            for (int i = 0; i < O_IOLEN - 1; ++i) in[i] = in[i + 1];
            in[O_IOLEN - 1] = W;
        }
        int all_eq = 1;
        for (int i = 0; i < O_IOLEN; ++i) all_eq &= (out[i] == tout[i]);
        /* :340-345  `(execute_jump & c > ADDRESS_MAX)` parses as `(execute_jump & c) > ADDRESS_MAX`
         * (Python: & binds tighter than >), i.e. (jump & c) in {0,1} compared with W-4: dead for W >= 16. */
        halt = (((jump ? 1 : 0) & c) > AMAX) || all_eq;
        /* :346-350; out_cursor-1 == -1 wraps to the last element in jnp, irrelevant because modified==0 then */
        int oc = out_cursor - 1;
        if (oc < 0) oc += O_IOLEN;
        error = err_a || err_b || werr || (modified && (out[oc] != tout[oc]));
    }
    memcpy(res->in_after, in, sizeof in);
    memcpy(res->out_after, out, sizeof out);
    res->bytes_used = bytes_used;
    res->cycles = cycles;
    int all_eq = 1;
    for (int i = 0; i < O_IOLEN; ++i) all_eq &= (out[i] == tout[i]);
    res->correct = !error && all_eq; /* :394 */
}

/* src/envs/subleq.py:504-532 (run_tests) + the reward function (:535-537 solved_or_not / :540-542 lowest_bytes);
 * writes the fields _step/_init replace */
static void o_subleq_run_tests(int W, int reward_fn, OSubleqState* s) {
    int32_t tin[3][O_IOLEN], tout[3][O_IOLEN];
    o_subleq_test_cases(s->task, W, tin, tout);
    int solved = 1, bytes_used = 0;
    OSimResult r0;
    for (int c = 0; c < 3; ++c) {
        OSimResult r;
        o_subleq_simulate(W, s->memory, tin[c], tout[c], &r);
        solved &= r.correct;
        if (r.bytes_used > bytes_used) bytes_used = r.bytes_used; /* :530 results.bytes_used.max() */
        if (c == 0) r0 = r;
    }
    memcpy(s->ex_in, tin[0], sizeof tin[0]);
    memcpy(s->ex_out, tout[0], sizeof tout[0]);
    memcpy(s->ex_in_after, r0.in_after, sizeof r0.in_after);
    memcpy(s->ex_out_after, r0.out_after, sizeof r0.out_after);
    s->solved = (uint8_t)solved;
    /* solved.astype(f32) / (1 + bytes_used): int32 + float32 promote to float32, one correctly rounded division */
    s->reward = reward_fn ? (solved ? 1.0f : 0.0f) / (float)(1 + bytes_used) : (solved ? 1.0f : 0.0f);
}

/* src/envs/subleq.py:623-646 (_init).  `jax.random.choice(key, tasks)` -> tasks[bits % n] with the build's RNG. */
static void o_subleq_init(const OEnv* e, OKey key, OSubleqState* s) {
    memset(s, 0, sizeof(*s));
    int idx = e->n_tasks > 1 ? (int)(o_bits(key, 0) % (uint32_t)e->n_tasks) : 0;
    s->task = e->tasks[idx];
    o_subleq_run_tests(e->W, e->reward_fn, s);
}

/* src/envs/subleq.py:648-677 (_step); step_count already incremented */
static void o_subleq_step(const OEnv* e, OSubleqState* s, int action) {
    const int W = e->W;
    if (s->step_count >= W - 3 || s->solved) { /* :671-673 */
        s->terminated = 1;
        s->reward = 0.0f;
        return;
    }
    int idx = s->step_count - 1; /* :654 */
    if (idx >= 0 && idx < W) s->memory[idx] = action;
    o_subleq_run_tests(W, e->reward_fn, s);
}

/* src/envs/subleq.py:26-56 (one-hot) and :63-98 (binary) encoders, one row */
static void o_subleq_encode_row(const OEnv* e, int32_t v, uint8_t* row) {
    const int W = e->W, C = e->obs_cols;
    memset(row, 0, (size_t)C);
    if (e->binary) {
        unsigned m = (unsigned)o_pymod(v, W) & 0xFFu; /* astype(uint8) :90 */
        for (int j = 0; j < C; ++j) row[j] = j < 8 ? (uint8_t)((m >> j) & 1u) : 0; /* unpackbits little, count=C :91 */
        row[C - 1] = (uint8_t)(v == W); /* :96 */
    } else {
        int idx = v == W ? W : o_pymod(v, W); /* :52 */
        row[idx] = 1;
    }
}

/* src/envs/subleq.py:679-707 (_observe): memory, example_input, input_after, example_output, output_after */
static void o_subleq_observe(const OEnv* e, const OSubleqState* s, uint8_t* obs) {
    const int W = e->W, C = e->obs_cols;
    uint8_t* p = obs;
    for (int i = 0; i < W; ++i, p += C) o_subleq_encode_row(e, s->memory[i], p);
    for (int i = 0; i < O_IOLEN; ++i, p += C) o_subleq_encode_row(e, s->ex_in[i], p);
    for (int i = 0; i < O_IOLEN; ++i, p += C) o_subleq_encode_row(e, s->ex_in_after[i], p);
    for (int i = 0; i < O_IOLEN; ++i, p += C) o_subleq_encode_row(e, s->ex_out[i], p);
    for (int i = 0; i < O_IOLEN; ++i, p += C) o_subleq_encode_row(e, s->ex_out_after[i], p);
}

/* ---- connect_four: pgx.connect_four [EXT] -- **parity unpinned** (pgx is un-vendored, Pipfile:8 "*").  Restated from the
 * public pgx v2 game: 6x7 board stored row-major with row 0 at the TOP, stones fall to the highest free row index,
 * color 0 moves first, a player identity (current_player) is drawn at init, winner = +1 / loser = -1, draw = 0,
 * observation [6,7,2] = (stones of the player to move, stones of the opponent).  Used by BASELINE.json configs[3]. ---- */
typedef struct {
    int32_t step_count;      /* _step_count */
    int32_t current_player;  /* State.current_player (identity of the player to move) */
    float reward0;           /* rewards[0] */
    uint8_t terminated, truncated, color /* GameState.color: whose stones are placed next */, pad;
    float reward1;           /* rewards[1] */
    int32_t winner;          /* GameState.winner: -1 none, else color */
    int8_t board[42];        /* -1 empty, 0 / 1 stone color */
    uint8_t pad2[14];
} OConnect4State; /* 80 bytes, identical to eaz_connect4_state of include/eaz.h */

static void o_c4_init(OKey key, OConnect4State* s) {
    memset(s, 0, sizeof(*s));
    s->current_player = (int32_t)(o_bits(key, 0) & 1u); /* jnp.int32(jax.random.bernoulli(key)) */
    s->winner = -1;
    for (int i = 0; i < 42; ++i) s->board[i] = -1;
}
static int o_c4_col_count(const OConnect4State* s, int c) {
    int n = 0;
    for (int r = 0; r < 6; ++r) n += s->board[r * 7 + c] >= 0;
    return n;
}
static int o_c4_won(const int8_t* b, int color) { /* any of the 69 lines of four */
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 7; ++c) {
            static const int D[4][2] = {{0, 1}, {1, 0}, {1, 1}, {1, -1}};
            for (int d = 0; d < 4; ++d) {
                int ok = 1;
                for (int k = 0; k < 4 && ok; ++k) {
                    const int rr = r + k * D[d][0], cc = c + k * D[d][1];
                    ok = rr >= 0 && rr < 6 && cc >= 0 && cc < 7 && b[rr * 7 + cc] == color;
                }
                if (ok) return 1;
            }
        }
    return 0;
}
/* ConnectFour._step (step_count already incremented) */
static void o_c4_step(OConnect4State* s, int action) {
    const int n = o_c4_col_count(s, action);
    const int color = s->color;
    const int mover = s->current_player;
    if (n < 6) s->board[(5 - n) * 7 + action] = (int8_t)color; /* a full column is an illegal action: handled by the wrapper */
    const int won = o_c4_won(s->board, color);
    s->winner = won ? color : -1;
    s->color = (uint8_t)(1 - color);
    s->current_player = 1 - mover;
    int full = 1;
    for (int c = 0; c < 7; ++c) full &= o_c4_col_count(s, c) == 6;
    s->terminated = (uint8_t)(won || full);
    /* rewards: the player who completed four gets +1, the other -1; zeros otherwise */
    s->reward0 = 0.0f; s->reward1 = 0.0f;
    if (won) { if (mover == 0) { s->reward0 = 1.0f; s->reward1 = -1.0f; } else { s->reward0 = -1.0f; s->reward1 = 1.0f; } }
}
/* observation from the perspective of the player to move: [6,7,2] = (board == my color, board == other color) */
static void o_c4_observe(const OConnect4State* s, uint8_t* obs) {
    for (int i = 0; i < 42; ++i) {
        obs[2 * i] = (uint8_t)(s->board[i] == (int8_t)s->color);
        obs[2 * i + 1] = (uint8_t)(s->board[i] == (int8_t)(1 - s->color));
    }
}

/* ---- generic env (pgx.Env wrapper [EXT]; called at context.py:127, selfplay.py:135,166) - */
/* every state blob starts with {int32 step_count; int32 aux; float reward0; uint8 terminated, truncated, ..} */
typedef struct { int32_t step_count, aux; float reward0; uint8_t terminated, truncated, x, y; } OStateHeader;

static int o_state_terminated(const OEnv* e, const void* s) { (void)e; return ((const OStateHeader*)s)->terminated; }
static int o_state_truncated(const OEnv* e, const void* s) { (void)e; return ((const OStateHeader*)s)->truncated; }
static int o_state_current_player(const OEnv* e, const void* s) {
    return e->kind == O_ENV_CONNECT4 ? ((const OConnect4State*)s)->current_player : 0;
}
/* rewards[player] */
static float o_state_reward(const OEnv* e, const void* s, int player) {
    if (e->kind == O_ENV_CONNECT4 && player == 1) return ((const OConnect4State*)s)->reward1;
    return ((const OStateHeader*)s)->reward0;
}
static void o_state_zero_rewards(const OEnv* e, void* s) {
    ((OStateHeader*)s)->reward0 = 0.0f;
    if (e->kind == O_ENV_CONNECT4) ((OConnect4State*)s)->reward1 = 0.0f;
}
static void o_state_set_terminated(const OEnv* e, void* s) { (void)e; ((OStateHeader*)s)->terminated = 1; }
/* legal_action_mask: all-true for DeepSea (deep_sea.py:19) and Subleq (subleq.py:626); connect_four: column not full.
 * pgx resets the mask to all-true at terminal states. */
static int o_state_legal(const OEnv* e, const void* s, int a) {
    if (e->kind != O_ENV_CONNECT4) return 1;
    const OConnect4State* c = (const OConnect4State*)s;
    if (c->terminated) return 1;
    return o_c4_col_count(c, a) < 6;
}

static void o_env_init(const OEnv* e, OKey key, void* s) {
    if (e->kind == O_ENV_DEEPSEA) o_deepsea_init(e, (ODeepSeaState*)s);
    else if (e->kind == O_ENV_SUBLEQ) o_subleq_init(e, key, (OSubleqState*)s);
    else o_c4_init(key, (OConnect4State*)s);
}

/* pgx.Env.step [EXT] */
static void o_env_step(const OEnv* e, const void* in, int action, OKey key, void* out) {
    (void)key;
    int illegal = !o_state_legal(e, in, action);
    memcpy(out, in, (size_t)e->state_bytes);
    const int mover = o_state_current_player(e, in);
    if (o_state_terminated(e, in) || o_state_truncated(e, in)) {
        o_state_zero_rewards(e, out); /* same state, zero rewards */
    } else if (e->kind == O_ENV_DEEPSEA) {
        ODeepSeaState* s = (ODeepSeaState*)out;
        s->step_count += 1;
        o_deepsea_step(e, s, action);
    } else if (e->kind == O_ENV_SUBLEQ) {
        OSubleqState* s = (OSubleqState*)out;
        s->step_count += 1;
        o_subleq_step(e, s, action);
    } else {
        OConnect4State* s = (OConnect4State*)out;
        s->step_count += 1;
        o_c4_step(s, action);
    }
    if (illegal) { /* _step_with_illegal_action: the mover gets -1, every other player +1 * (num_players - 1); terminated */
        ((OStateHeader*)out)->reward0 = -1.0f;
        if (e->kind == O_ENV_CONNECT4) {
            OConnect4State* s = (OConnect4State*)out;
            s->reward0 = mover == 0 ? -1.0f : 1.0f;
            s->reward1 = mover == 1 ? -1.0f : 1.0f;
        }
        o_state_set_terminated(e, out);
    }
}

static void o_env_observe(const OEnv* e, const void* s, uint8_t* obs) {
    if (e->kind == O_ENV_DEEPSEA) o_deepsea_observe(e, (const ODeepSeaState*)s, obs);
    else if (e->kind == O_ENV_SUBLEQ) o_subleq_observe(e, (const OSubleqState*)s, obs);
    else o_c4_observe((const OConnect4State*)s, obs);
}

/* src/selfplay.py:26-75 (auto_reset): reset if ALREADY terminal, else step.  key1 -> init, key2 -> step */
static void o_env_step_autoreset(const OEnv* e, const void* in, int action, OKey key, void* out) {
    if (o_state_terminated(e, in) || o_state_truncated(e, in)) o_env_init(e, o_fold(key, 1), out);
    else o_env_step(e, in, action, o_fold(key, 2), out);
}

/* ---- exported env API ------------------------------------------------------------------------------- */

EXPORT OEnv* eazo_env_deepsea(int G, const uint8_t* action_map) {
    OEnv* e = (OEnv*)calloc(1, sizeof(OEnv));
    e->kind = O_ENV_DEEPSEA;
    e->G = G;
    e->num_actions = 2;
    e->obs_rows = G;
    e->obs_cols = G;
    e->state_bytes = (int)sizeof(ODeepSeaState);
    e->action_map = (uint8_t*)calloc((size_t)G * G, 1);
    if (action_map) memcpy(e->action_map, action_map, (size_t)G * G);
    return e;
}

EXPORT OEnv* eazo_env_subleq(int W, const int* tasks, int n_tasks, int binary) {
    if (W < 16 || W > 256 || n_tasks < 1 || n_tasks > 16) return NULL; /* subleq.py:606 */
    OEnv* e = (OEnv*)calloc(1, sizeof(OEnv));
    e->kind = O_ENV_SUBLEQ;
    e->W = W;
    e->num_actions = W;
    e->n_tasks = n_tasks;
    for (int i = 0; i < n_tasks; ++i) e->tasks[i] = tasks[i];
    e->binary = binary;
    int bl = 0;
    for (int v = W - 1; v > 0; v >>= 1) bl++;
    e->enc_width = bl + 1;                      /* subleq.py:59-60 */
    e->obs_rows = W + 4 * O_IOLEN;              /* subleq.py:618-621 */
    e->obs_cols = binary ? e->enc_width : W + 1;
    e->state_bytes = (int)sizeof(OSubleqState);
    return e;
}

/* Subleq(reward_fn=...) subleq.py:603: 0 = solved_or_not, 1 = lowest_bytes */
EXPORT int eazo_env_set_subleq_reward(OEnv* e, int reward_fn) {
    if (!e || e->kind != O_ENV_SUBLEQ || reward_fn < 0 || reward_fn > 1) return -1;
    e->reward_fn = reward_fn;
    return 0;
}

EXPORT OEnv* eazo_env_connect4(void) {
    OEnv* e = (OEnv*)calloc(1, sizeof(OEnv));
    e->kind = O_ENV_CONNECT4;
    e->num_actions = 7;
    e->obs_rows = 42; /* [6,7,2] flattened as 42 cells x 2 planes */
    e->obs_cols = 2;
    e->state_bytes = (int)sizeof(OConnect4State);
    return e;
}

EXPORT void eazo_env_free(OEnv* e) {
    if (!e) return;
    free(e->action_map);
    free(e);
}
EXPORT int eazo_env_state_bytes(const OEnv* e) { return e->state_bytes; }
EXPORT int eazo_env_num_actions(const OEnv* e) { return e->num_actions; }
EXPORT int eazo_env_obs_dim(const OEnv* e) { return e->obs_rows * e->obs_cols; }
EXPORT int eazo_env_obs_rows(const OEnv* e) { return e->obs_rows; }
EXPORT int eazo_env_obs_cols(const OEnv* e) { return e->obs_cols; }

EXPORT void eazo_env_init(const OEnv* e, int B, uint32_t k0, uint32_t k1, void* states) {
    OKey key = {k0, k1};
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) o_env_init(e, o_fold(key, (uint32_t)b), (char*)states + (size_t)b * e->state_bytes);
}

/* vmap(env.step) or vmap(auto_reset(env.step, env.init)) */
EXPORT void eazo_env_step(const OEnv* e, int B, const void* in, const int32_t* action, uint32_t k0, uint32_t k1,
                          int autoreset, void* out) {
    OKey key = {k0, k1};
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const void* si = (const char*)in + (size_t)b * e->state_bytes;
        void* so = (char*)out + (size_t)b * e->state_bytes;
        if (autoreset) o_env_step_autoreset(e, si, action[b], o_fold(key, (uint32_t)b), so);
        else o_env_step(e, si, action[b], o_fold(key, (uint32_t)b), so);
    }
}

EXPORT void eazo_env_observe(const OEnv* e, int B, const void* states, uint8_t* obs) {
    const int D = e->obs_rows * e->obs_cols;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) o_env_observe(e, (const char*)states + (size_t)b * e->state_bytes, obs + (size_t)b * D);
}

/* unpack the pgx.State-visible fields: terminated, truncated, rewards[:,0], _step_count */
EXPORT void eazo_env_fields(const OEnv* e, int B, const void* states, uint8_t* terminated, uint8_t* truncated,
                            float* reward, int32_t* step_count) {
    for (int b = 0; b < B; ++b) {
        const void* s = (const char*)states + (size_t)b * e->state_bytes;
        if (terminated) terminated[b] = (uint8_t)o_state_terminated(e, s);
        if (truncated) truncated[b] = (uint8_t)o_state_truncated(e, s);
        if (reward) reward[b] = o_state_reward(e, s, 0);
        if (step_count) step_count[b] = ((const OStateHeader*)s)->step_count;
    }
}

/* raw access used by the Subleq known-answer tests */
EXPORT void eazo_subleq_test_cases(int task, int W, int32_t* tin, int32_t* tout) {
    int32_t a[3][O_IOLEN], b[3][O_IOLEN];
    o_subleq_test_cases(task, W, a, b);
    memcpy(tin, a, sizeof a);
    memcpy(tout, b, sizeof b);
}
EXPORT void eazo_subleq_simulate(int W, const int32_t* mem, const int32_t* tin, const int32_t* tout, int32_t* in_after,
                                 int32_t* out_after, int32_t* info /* bytes, cycles, correct */) {
    OSimResult r;
    o_subleq_simulate(W, mem, tin, tout, &r);
    memcpy(in_after, r.in_after, sizeof r.in_after);
    memcpy(out_after, r.out_after, sizeof r.out_after);
    info[0] = r.bytes_used; info[1] = r.cycles; info[2] = r.correct;
}
EXPORT void eazo_subleq_encode(const OEnv* e, const int32_t* words, int n, uint8_t* rows) {
    for (int i = 0; i < n; ++i) o_subleq_encode_row(e, words[i], rows + (size_t)i * e->obs_cols);
}
/* encoders for ANY word size (the docstring example at subleq.py:30-41 uses word_size = 8, below the env's minimum) */
EXPORT int eazo_subleq_encode_raw(int W, int binary, const int32_t* words, int n, uint8_t* rows) {
    OEnv e;
    memset(&e, 0, sizeof e);
    e.W = W; e.binary = binary;
    int bl = 0;
    for (int v = W - 1; v > 0; v >>= 1) bl++;
    e.obs_cols = binary ? bl + 1 : W + 1;
    for (int i = 0; i < n; ++i) o_subleq_encode_row(&e, words[i], rows + (size_t)i * e.obs_cols);
    return e.obs_cols;
}

/* ===================================================================================================== *
 *  Hash novelty set  (src/network/hashes.py)                                                            *
 * ===================================================================================================== */

/* XXHash.get_indices, src/network/hashes.py:162-229, on the f32-bitcast of a byte observation.
 * NOT real xxHash32: lane i is the i-th contiguous quarter (:211), seed 1 (:169), "+ L" stripes (:226). */
static uint32_t o_xxhash_variant(const uint8_t* x, int n /* multiple of 4 */, int bits_per_hash) {
    const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, SEED = 1u;
    const int L = n / 4;
    uint32_t acc[4] = {SEED + P1 + P2, SEED + P2, SEED + 0u, SEED - P1}; /* :217-220 */
    for (int l = 0; l < L; ++l)
        for (int i = 0; i < 4; ++i) {
            uint32_t lane = o_f2u((float)x[i * L + l]); /* astype(f32) fully_connected.py:45 ; bitcast hashes.py:207 */
            uint32_t a = acc[i] + lane * P2;              /* :176-181 */
            a = o_rotl32(a, 13);
            acc[i] = a * P1;
        }
    uint32_t h = o_rotl32(acc[0], 1) + o_rotl32(acc[1], 7) + o_rotl32(acc[2], 12) + o_rotl32(acc[3], 18); /* :184-187 */
    h += (uint32_t)L;                                                                                   /* :226 */
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;                                         /* :190-197 */
    return h >> (32 - bits_per_hash);                                                                   /* :229 */
}

/* BaseHash.__call__, src/network/hashes.py:23-38 */
/* SimHash.get_indices, src/network/hashes.py:124-139: product = x @ random_matrix [n, bits]; index = sum_j (product_j < 0) << j.
 * Spec of the build for the product (XLA leaves the order of a dot unspecified): acc = +0; acc = fmaf(x[k], R[k][j], acc), k ascending
 * -- the hk.Linear chain (o_linear). */
static uint32_t o_simhash(const uint8_t* x, int n, int bits, const float* R) {
    uint32_t idx = 0;
    for (int j = 0; j < bits; ++j) {
        float acc = 0.0f;
        for (int k = 0; k < n; ++k) acc = fmaf((float)x[k], R[(size_t)k * bits + j], acc);
        if (acc < 0.0f) idx |= 1u << j;
    }
    return idx;
}
EXPORT void eazo_simhash_indices(const uint8_t* x, int B, int n, int bits, const float* R, uint32_t* out) {
    for (int b = 0; b < B; ++b) out[b] = o_simhash(x + (size_t)b * n, n, bits, R);
}

static int o_hash_seen(const uint8_t* set, uint32_t index) { return (set[index >> 3] & (1u << (index & 7u))) != 0; }

EXPORT void eazo_xxhash_indices(const uint8_t* x, int B, int n, int bits, uint32_t* out) {
    for (int b = 0; b < B; ++b) out[b] = o_xxhash_variant(x + (size_t)b * n, n, bits);
}
/* BaseHash.update, src/network/hashes.py:45-50 (scatter-OR; "next" row N3) */
EXPORT void eazo_hash_update(uint8_t* set, const uint32_t* idx, int B) {
    for (int b = 0; b < B; ++b) set[idx[b] >> 3] |= (uint8_t)(1u << (idx[b] & 7u));
}
EXPORT void eazo_hash_seen(const uint8_t* set, const uint32_t* idx, int B, uint8_t* seen) {
    for (int b = 0; b < B; ++b) seen[b] = (uint8_t)o_hash_seen(set, idx[b]);
}

/* primitives exported so the test can rebuild REAL xxh32 from them and pin them against the `xxhash` package */
EXPORT uint32_t eazo_xxh_round(uint32_t acc, uint32_t lane) {
    acc += lane * 0x85EBCA77u; acc = o_rotl32(acc, 13); return acc * 0x9E3779B1u;
}
EXPORT uint32_t eazo_xxh_converge(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
    return o_rotl32(a0, 1) + o_rotl32(a1, 7) + o_rotl32(a2, 12) + o_rotl32(a3, 18);
}
EXPORT uint32_t eazo_xxh_avalanche(uint32_t h) {
    h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13; h *= 0xC2B2AE3Du; h ^= h >> 16; return h;
}

/* ===================================================================================================== *
 *  Fully-connected epistemic net (src/network/fully_connected.py:41-101)                                *
 * ===================================================================================================== */

struct OResNet;
typedef struct ONet {
    struct OResNet* resnet; /* non-NULL: EpistemicResidualAZNet (src/network/resnet.py), the FC fields below are unused */
    int in_dim, H, A;
    float* w[12]; /* haiku creation order: value 0-2, ube 3-5, exploitation 6-8, exploration 9-11; w:[in,out] */
    float* b[12];
    float max_u;     /* fully_connected.py:19 (context.py:68-75 never overrides -> 1.0) */
    float nov_scale; /* max_epistemic_variance_reward, :20 */
    int hash_off, hash_len, hash_bits;
    const uint8_t* bitset; /* 2^(bits-3) bytes, caller-owned */
    const float* sim_matrix; /* non-NULL: hash_class = SimHash with this [hash_len, hash_bits] random matrix (caller-owned); NULL: XXHash */
    int* order;      /* non-NULL: layer-1 input order, a permutation of [0, in_dim) (eazo_net_set_input_order) */
    float* w1p[4];   /* the heads' layer-1 weights with their rows in that order */
} ONet;

/* hash_obj.get_indices(hash_input) of the configured hash class (config.py:27, context.py:42-49) */
static uint32_t o_net_hash_index(const ONet* n, const uint8_t* x) {
    return n->sim_matrix ? o_simhash(x, n->hash_len, n->hash_bits, n->sim_matrix) : o_xxhash_variant(x, n->hash_len, n->hash_bits);
}
EXPORT void eazo_net_set_simhash(ONet* n, const float* R) { n->sim_matrix = R; }

static int o_layer_in(const ONet* n, int l) { return (l % 3) == 0 ? n->in_dim : n->H; }
static int o_layer_out(const ONet* n, int l) { return (l % 3) < 2 ? n->H : ((l / 3) < 2 ? 1 : n->A); }

EXPORT ONet* eazo_net_create(int in_dim, int H, int A, const float* const* w, const float* const* b, float max_u,
                             float nov_scale, int hash_off, int hash_len, int hash_bits, const uint8_t* bitset) {
    ONet* n = (ONet*)calloc(1, sizeof(ONet));
    n->in_dim = in_dim; n->H = H; n->A = A;
    for (int l = 0; l < 12; ++l) {
        size_t ki = (size_t)o_layer_in(n, l), ko = (size_t)o_layer_out(n, l);
        n->w[l] = (float*)malloc(ki * ko * sizeof(float));
        n->b[l] = (float*)malloc(ko * sizeof(float));
        memcpy(n->w[l], w[l], ki * ko * sizeof(float));
        memcpy(n->b[l], b[l], ko * sizeof(float));
    }
    n->max_u = max_u; n->nov_scale = nov_scale;
    n->hash_off = hash_off; n->hash_len = hash_len; n->hash_bits = hash_bits; n->bitset = bitset;
    n->sim_matrix = NULL;
    return n;
}
static void o_resnet_free(struct OResNet* r);
EXPORT void eazo_net_free(ONet* n) {
    if (!n) return;
    if (n->resnet) { o_resnet_free(n->resnet); free(n); return; }
    for (int l = 0; l < 12; ++l) { free(n->w[l]); free(n->b[l]); }
    free(n->order);
    for (int h = 0; h < 4; ++h) free(n->w1p[h]);
    free(n);
}
EXPORT void eazo_net_set_bitset(ONet* n, const uint8_t* bitset) { n->bitset = bitset; }

/* Layer-1 INPUT ORDER (a switch of the build's numerics contract; no reference counterpart: XLA leaves the order of a dot
 * unspecified).  By default the layer-1 chain of o_linear visits the inputs k ascending.  With an order -- n_seg segments (start, len) whose
 * concatenation is a permutation of [0, in_dim) -- the chain visits the inputs in that order instead: acc = +0; for s in segments: for k in
 * start_s .. start_s + len_s - 1: acc = fmaf(x[k], w[k][j], acc); y = acc + b[j].  Subleq uses it to put the inputs that are constants of
 * an episode (the example_input / example_output rows, subleq.py:398-501,697-705) first, so that a device kernel may start every chain of a
 * tree from one precomputed partial sum.  Returns 0, or -1 if the segments are not a permutation.  n_seg == 0 restores the default. */
EXPORT int eazo_net_set_input_order(ONet* n, int n_seg, const int32_t* start, const int32_t* len) {
    if (n->resnet) return -1;
    free(n->order); n->order = NULL;
    for (int h = 0; h < 4; ++h) { free(n->w1p[h]); n->w1p[h] = NULL; }
    if (n_seg == 0) return 0;
    const int K = n->in_dim, H = n->H;
    int* ord = (int*)malloc((size_t)K * sizeof(int));
    uint8_t* seen = (uint8_t*)calloc((size_t)K, 1);
    int pos = 0, ok = 1;
    for (int s = 0; s < n_seg && ok; ++s)
        for (int k = start[s]; k < start[s] + len[s]; ++k) {
            if (k < 0 || k >= K || seen[k] || pos >= K) { ok = 0; break; }
            seen[k] = 1; ord[pos++] = k;
        }
    free(seen);
    if (!ok || pos != K) { free(ord); return -1; }
    n->order = ord;
    for (int h = 0; h < 4; ++h) {
        n->w1p[h] = (float*)malloc((size_t)K * H * sizeof(float));
        for (int kp = 0; kp < K; ++kp) memcpy(n->w1p[h] + (size_t)kp * H, n->w[h * 3] + (size_t)ord[kp] * H, (size_t)H * sizeof(float));
    }
    return 0;
}

/* hk.Linear: y = x @ w + b.  Spec of the build: acc = +0; acc = fmaf(x[k], w[k][j], acc) for k ascending; y = acc + b[j].
 * (This is exactly what one chain of gfx950 f32 MFMA accumulation computes.)  Rows are blocked by 4 so the j loop vectorises.
 *
 * parts > 1 (the THIRD layer of every FC head, K = linear_layer_size): the k range is cut into `parts` equal contiguous pieces,
 * each piece is its own fmaf chain from +0, and the pieces are added pairwise in the fixed order ((c0 + c1) + (c2 + c3)) before the
 * bias -- the four wavefronts of a device workgroup own a quarter of the hidden columns each and every one runs the chain over its
 * own quarter (DESIGN.md "Numerics contract").  XLA does not specify the reduction order of a dot at all; either
 * order is within 1e-5 of the float64 restatement (tests/test_oracle_hash_net.py). */
static void o_linear_parts(int B, int K, int N, const float* x, const float* w, const float* bias, float* y, int relu, int parts) {
    if (parts < 1 || K % parts != 0 || (parts != 1 && parts != 4)) parts = 1;
    const int Kp = K / parts;
#pragma omp parallel for schedule(static)
    for (int r0 = 0; r0 < B; r0 += 4) {
        float acc[4][4][O_MAX_A > 1024 ? O_MAX_A : 1024];  /* N = linear_layer_size (<= 1024) or the number of actions */
        const int rb = B - r0 < 4 ? B - r0 : 4;
        for (int q = 0; q < parts; ++q) {
            for (int r = 0; r < rb; ++r)
                for (int j = 0; j < N; ++j) acc[q][r][j] = 0.0f;
            for (int k = q * Kp; k < (q + 1) * Kp; ++k) {
                const float* wk = w + (size_t)k * N;
                for (int r = 0; r < rb; ++r) {
                    const float xv = x[(size_t)(r0 + r) * K + k];
                    float* ar = acc[q][r];
                    for (int j = 0; j < N; ++j) ar[j] = fmaf(xv, wk[j], ar[j]);
                }
            }
        }
        for (int r = 0; r < rb; ++r)
            for (int j = 0; j < N; ++j) {
                float v = parts == 4 ? (acc[0][r][j] + acc[1][r][j]) + (acc[2][r][j] + acc[3][r][j]) : acc[0][r][j];
                v = v + bias[j];
                if (relu) v = v > 0.0f ? v : 0.0f;
                y[(size_t)(r0 + r) * N + j] = v;
            }
    }
}
static void o_linear(int B, int K, int N, const float* x, const float* w, const float* bias, float* y, int relu) {
    o_linear_parts(B, K, N, x, w, bias, y, relu, 1);
}

/* EpistemicFullyConnectedAZNet.__call__(x, is_training=False), src/network/fully_connected.py:41-101.
 * Returns (exploitation_logits, exploration_logits, v, u, scaled_state_novelty) -- the code's order (:101). */
static void o_resnet_forward(const ONet* n, int B, const uint8_t* obs, float* exploit, float* explore, float* v, float* u, float* nov);
EXPORT void eazo_net_forward(const ONet* n, int B, const uint8_t* obs, float* exploit, float* explore, float* v,
                             float* u, float* nov) {
    if (n->resnet) { o_resnet_forward(n, B, obs, exploit, explore, v, u, nov); return; }
    const int K = n->in_dim, H = n->H, A = n->A;
    if (H > 1024 || A > O_MAX_A) { fprintf(stderr, "eazo_net_forward: H/A too large\n"); abort(); }
    float* x = (float*)malloc((size_t)B * K * sizeof(float));
    float* h1 = (float*)malloc((size_t)B * H * sizeof(float));
    float* h2 = (float*)malloc((size_t)B * H * sizeof(float));
    float* o = (float*)malloc((size_t)B * (A > 1 ? A : 1) * sizeof(float));
    if (n->order) {  /* the layer-1 chain visits the inputs in the configured order (eazo_net_set_input_order): same products, same bias */
        for (int b = 0; b < B; ++b)
            for (int kp = 0; kp < K; ++kp) x[(size_t)b * K + kp] = (float)obs[(size_t)b * K + n->order[kp]];
    } else {
        for (size_t i = 0; i < (size_t)B * K; ++i) x[i] = (float)obs[i]; /* :45 astype(float32) */
    }
    for (int head = 0; head < 4; ++head) {
        const int l = head * 3;
        const int No = o_layer_out(n, l + 2);
        o_linear(B, K, H, x, n->order ? n->w1p[head] : n->w[l], n->b[l], h1, 1);
        o_linear(B, H, H, h1, n->w[l + 1], n->b[l + 1], h2, 1);
        o_linear_parts(B, H, No, h2, n->w[l + 2], n->b[l + 2], o, 0, 4);  /* four partial chains, see o_linear_parts */
        if (head == 0) { for (int b = 0; b < B; ++b) v[b] = o_tanhf(o[b]); }                       /* :50-56 */
        else if (head == 1) { for (int b = 0; b < B; ++b) u[b] = 0.5f * (o_tanhf(o[b]) + 1.0f); }  /* :59-65 */
        else memcpy(head == 2 ? exploit : explore, o, (size_t)B * A * sizeof(float));             /* :68-81 */
    }
    for (int b = 0; b < B; ++b) {
        /* :83-90 novelty = ~seen * max_reward_epistemic_variance */
        int seen = 0;
        if (n->bitset) seen = o_hash_seen(n->bitset, o_net_hash_index(n, obs + (size_t)b * K + n->hash_off));
        float novelty = (seen ? 0.0f : 1.0f) * n->nov_scale;
        /* :92-96 (not is_training): u = clip(max(novelty, u*max_u), 0, max_u) */
        float uu = u[b] * n->max_u;
        uu = novelty > uu ? novelty : uu;
        uu = uu < 0.0f ? 0.0f : uu;
        uu = uu > n->max_u ? n->max_u : uu;
        u[b] = uu;
        if (nov) nov[b] = novelty;
    }
    free(x); free(h1); free(h2); free(o);
}

/* ===================================================================================================== *
 *  Residual epistemic net (src/network/resnet.py:42-135), NHWC, BlockV2 (:26-39), eval-mode BatchNorm      *
 * ===================================================================================================== */
/* Numerics spec of the build (DESIGN.md "ResNet forward"):
 *   hk.Conv2D (SAME, stride 1): acc = +0; for tap (kh, kw) row-major, for ci ascending: acc = fmaf(x[h+kh-1][w+kw-1][ci], w[tap][ci][co], acc)
 *                               with out-of-board taps skipped (their contribution is an exact +-0); y = acc + b[co]
 *   hk.BatchNorm eval:          y = (x - mean) * inv + offset,  inv = scale / sqrtf(var + 1e-5f)  (precomputed per channel)
 *   hk.Linear:                  as in the FC net (k-ascending fmaf chain, then + b) */
typedef struct { float *w, *b; int taps, cin, cout; } OConv;
typedef struct { float *scale, *offset, *mean, *var, *inv; int c; } OBN;
typedef struct { float *w, *b; int in, out; } OLin;
typedef struct OResNet {
    int H, W, Cin, C, nb, A;
    OConv conv0, c1[16], c2[16];
    OBN bn1[16], bn2[16], bnf;
    OConv hc[4]; /* 1x1 convs of the policy, exploration, value, ube heads (resnet.py:84,92,102,114) */
    OBN hbn[4];
    OLin pl, xl, vl1, vl2, ul1, ul2;
} OResNet;

static float* o_dup(const float* p, size_t n) { float* q = (float*)malloc(n * sizeof(float)); memcpy(q, p, n * sizeof(float)); return q; }
static int o_take_conv(OConv* c, const float* const* a, int i, int taps, int cin, int cout) {
    c->taps = taps; c->cin = cin; c->cout = cout;
    c->w = o_dup(a[i], (size_t)taps * cin * cout); c->b = o_dup(a[i + 1], (size_t)cout);
    return i + 2;
}
static int o_take_bn(OBN* b, const float* const* a, int i, int c) {
    b->c = c;
    b->scale = o_dup(a[i], c); b->offset = o_dup(a[i + 1], c); b->mean = o_dup(a[i + 2], c); b->var = o_dup(a[i + 3], c);
    b->inv = (float*)malloc(sizeof(float) * c);
    for (int k = 0; k < c; ++k) b->inv[k] = b->scale[k] / sqrtf(b->var[k] + 1e-5f);
    return i + 4;
}
static int o_take_lin(OLin* l, const float* const* a, int i, int in, int out) {
    l->in = in; l->out = out;
    l->w = o_dup(a[i], (size_t)in * out); l->b = o_dup(a[i + 1], (size_t)out);
    return i + 2;
}
static void o_free_conv(OConv* c) { free(c->w); free(c->b); }
static void o_free_bn(OBN* b) { free(b->scale); free(b->offset); free(b->mean); free(b->var); free(b->inv); }
static void o_free_lin(OLin* l) { free(l->w); free(l->b); }
static void o_resnet_free(OResNet* r) {
    o_free_conv(&r->conv0); o_free_bn(&r->bnf);
    for (int i = 0; i < r->nb; ++i) { o_free_conv(&r->c1[i]); o_free_conv(&r->c2[i]); o_free_bn(&r->bn1[i]); o_free_bn(&r->bn2[i]); }
    for (int h = 0; h < 4; ++h) { o_free_conv(&r->hc[h]); o_free_bn(&r->hbn[h]); }
    o_free_lin(&r->pl); o_free_lin(&r->xl); o_free_lin(&r->vl1); o_free_lin(&r->vl2); o_free_lin(&r->ul1); o_free_lin(&r->ul2);
    free(r);
}

/* arrays in haiku creation order (resnet.py:72-121): conv0 {w,b}; per block {bn1 x4, conv1 {w,b}, bn2 x4, conv2 {w,b}}; final bn x4;
 * policy {conv w,b; bn x4; linear w,b}; exploration (same); value {conv; bn; linear(C) w,b; linear(1) w,b}; ube (same).
 * bn x4 = scale, offset, mean EMA, var EMA.  42 + 12*num_blocks arrays. */
EXPORT ONet* eazo_resnet_create(int H, int W, int Cin, int C, int num_blocks, int A, const float* const* a, int n_arrays,
                                int hash_bits, const uint8_t* bitset) {
    if (num_blocks > 16 || n_arrays != 42 + 12 * num_blocks || (H * W * Cin) % 4 != 0) return NULL;
    OResNet* r = (OResNet*)calloc(1, sizeof(OResNet));
    r->H = H; r->W = W; r->Cin = Cin; r->C = C; r->nb = num_blocks; r->A = A;
    int i = o_take_conv(&r->conv0, a, 0, 9, Cin, C);
    for (int k = 0; k < num_blocks; ++k) {
        i = o_take_bn(&r->bn1[k], a, i, C); i = o_take_conv(&r->c1[k], a, i, 9, C, C);
        i = o_take_bn(&r->bn2[k], a, i, C); i = o_take_conv(&r->c2[k], a, i, 9, C, C);
    }
    i = o_take_bn(&r->bnf, a, i, C);
    i = o_take_conv(&r->hc[0], a, i, 1, C, 2); i = o_take_bn(&r->hbn[0], a, i, 2); i = o_take_lin(&r->pl, a, i, H * W * 2, A);
    i = o_take_conv(&r->hc[1], a, i, 1, C, 2); i = o_take_bn(&r->hbn[1], a, i, 2); i = o_take_lin(&r->xl, a, i, H * W * 2, A);
    i = o_take_conv(&r->hc[2], a, i, 1, C, 1); i = o_take_bn(&r->hbn[2], a, i, 1); i = o_take_lin(&r->vl1, a, i, H * W, C); i = o_take_lin(&r->vl2, a, i, C, 1);
    i = o_take_conv(&r->hc[3], a, i, 1, C, 1); i = o_take_bn(&r->hbn[3], a, i, 1); i = o_take_lin(&r->ul1, a, i, H * W, C); i = o_take_lin(&r->ul2, a, i, C, 1);
    ONet* n = (ONet*)calloc(1, sizeof(ONet));
    n->resnet = r;
    n->in_dim = H * W * Cin; n->A = A; n->H = C;
    n->max_u = 1.0f; n->nov_scale = 1.0f; /* resnet.py:63-64 */
    n->hash_off = 0; n->hash_len = H * W * Cin; n->hash_bits = hash_bits; n->bitset = bitset;
    n->sim_matrix = NULL;
    return n;
}

/* y[p][co] = conv(x)[p][co] + b[co]   (x: [H*W][cin] NHWC) */
static void o_conv(const OResNet* r, const OConv* c, const float* x, float* y) {
    const int H = r->H, W = r->W, cin = c->cin, cout = c->cout, half = c->taps == 9 ? 1 : 0;
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            float acc[64];
            for (int co = 0; co < cout; ++co) acc[co] = 0.0f;
            for (int kh = -half; kh <= half; ++kh)
                for (int kw = -half; kw <= half; ++kw) {
                    const int hh = h + kh, ww = w + kw;
                    if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                    const int tap = (kh + half) * (2 * half + 1) + (kw + half);
                    const float* xp = x + (size_t)(hh * W + ww) * cin;
                    const float* wp = c->w + (size_t)tap * cin * cout;
                    for (int ci = 0; ci < cin; ++ci) {
                        const float xv = xp[ci];
                        for (int co = 0; co < cout; ++co) acc[co] = fmaf(xv, wp[(size_t)ci * cout + co], acc[co]);
                    }
                }
            for (int co = 0; co < cout; ++co) y[(size_t)(h * W + w) * cout + co] = acc[co] + c->b[co];
        }
}
static void o_bn_relu(const OBN* b, const float* x, float* y, int P) {
    for (int p = 0; p < P; ++p)
        for (int k = 0; k < b->c; ++k) {
            const float v = (x[(size_t)p * b->c + k] - b->mean[k]) * b->inv[k] + b->offset[k];
            y[(size_t)p * b->c + k] = v > 0.0f ? v : 0.0f;
        }
}
static void o_lin(const OLin* l, const float* x, float* y, int relu) {
    for (int j = 0; j < l->out; ++j) {
        float acc = 0.0f;
        for (int k = 0; k < l->in; ++k) acc = fmaf(x[k], l->w[(size_t)k * l->out + j], acc);
        acc = acc + l->b[j];
        y[j] = relu ? (acc > 0.0f ? acc : 0.0f) : acc;
    }
}

static void o_resnet_forward(const ONet* n, int B, const uint8_t* obs, float* exploit, float* explore, float* v, float* u, float* nov) {
    const OResNet* r = n->resnet;
    const int P = r->H * r->W, C = r->C, D = P * r->Cin, A = r->A;
    if (C > 64 || A > 64) { fprintf(stderr, "o_resnet_forward: C/A too large\n"); abort(); }
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        float x0[42 * 8], X[42 * 64], U[42 * 64], T[42 * 64], hbuf[42 * 2], hact[42 * 2], fc[64], out[64];
        for (int i = 0; i < D; ++i) x0[i] = (float)obs[(size_t)b * D + i];         /* resnet.py:69 astype(float32) */
        o_conv(r, &r->conv0, x0, X);                                                /* :70 */
        for (int k = 0; k < r->nb; ++k) {                                           /* BlockV2 :31-39 */
            o_bn_relu(&r->bn1[k], X, U, P);
            o_conv(r, &r->c1[k], U, T);
            o_bn_relu(&r->bn2[k], T, U, P);
            o_conv(r, &r->c2[k], U, T);
            for (int i = 0; i < P * C; ++i) X[i] = T[i] + X[i];                     /* :39 x + i */
        }
        o_bn_relu(&r->bnf, X, U, P);                                                /* :79-81 */
        for (int h = 0; h < 4; ++h) {
            o_conv(r, &r->hc[h], U, hbuf);
            o_bn_relu(&r->hbn[h], hbuf, hact, P);                                   /* flatten keeps (h, w, c) order */
            if (h == 0 && exploit) o_lin(&r->pl, hact, exploit + (size_t)b * A, 0);    /* :84-88 */
            else if (h == 1 && explore) o_lin(&r->xl, hact, explore + (size_t)b * A, 0); /* :91-98 */
            else if (h == 2) { o_lin(&r->vl1, hact, fc, 1); o_lin(&r->vl2, fc, out, 0); if (v) v[b] = o_tanhf(out[0]); }             /* :101-109 */
            else if (h == 3) { o_lin(&r->ul1, hact, fc, 1); o_lin(&r->ul2, fc, out, 0); if (u) u[b] = 0.5f * (o_tanhf(out[0]) + 1.0f); } /* :112-121 */
        }
        int seen = 0;
        if (n->bitset) seen = o_hash_seen(n->bitset, o_net_hash_index(n, obs + (size_t)b * D));
        const float novelty = (seen ? 0.0f : 1.0f) * n->nov_scale;                  /* :124-125 */
        if (u) u[b] = novelty > u[b] ? novelty : u[b];                              /* :127-130 (not is_training) */
        if (nov) nov[b] = novelty;
    }
}

/* ===================================================================================================== *
 *  Gumbel-MuZero search with epistemic fields  (emctx [EXT]; call site src/selfplay.py:100-121)         *
 *  **parity unpinned** -- restated from the public mctx design + E-MCTS paper, SURVEY.md Appendix A.     *
 * ===================================================================================================== */

typedef struct OTree {
    int B, N, A, S;
    int32_t *node_visits, *parents, *action_from_parent;                    /* [B,N] */
    float *raw_values, *node_values, *raw_var, *node_var;                   /* [B,N] */
    int32_t *children_index, *children_visits;                              /* [B,N,A] */
    float *children_prior_logits, *children_values, *children_rewards, *children_discounts; /* [B,N,A] */
    float *children_var, *children_reward_var;                              /* [B,N,A] epistemic twins */
    float* beta;           /* [B] */
    uint8_t* root_invalid; /* [B,A] */
    float* root_gumbel;    /* [B,A] */
    char* embeddings;      /* [B,N,state_bytes] */
    int state_bytes;
    /* per-simulation scratch */
    int32_t *sim_parent, *sim_action, *sim_next, *sim_player; /* [B] */
    uint8_t* obs;                                /* [B,obs_dim] */
    char* next_states;                           /* [B,state_bytes] */
    float *o_exploit, *o_explore, *o_v, *o_u, *o_nov;
    /* search options */
    const OEnv* env;
    const ONet* net;
    float discount;
    int two_player, use_exploration_logits, rescale, use_mixed, max_considered, max_depth;
    /* the two choices of the epistemic extension that cannot be checked against emctx offline (SURVEY.md A.5 / A.6 "(?)"), see
     * include/eaz.h eaz_search_opts: bonus_after_rescale 0 = beta * sigma added to the completed Q before the min-max rescale,
     * 1 = after it; var_backup 0 = running mean of the backed-up variances, 1 = variance of the running mean */
    int bonus_after_rescale, var_backup;
    float value_scale, maxvisit_init;
    int32_t* table; /* [(max_considered+1), S] sequential-halving considered-visit table */
} OTree;

/* mctx seq_halving.get_sequence_of_considered_visits [EXT] */
EXPORT void eazo_considered_visits(int m, int S, int32_t* out) {
    if (m <= 1) { for (int i = 0; i < S; ++i) out[i] = i; return; }
    int log2max = 0;
    while ((1 << log2max) < m) log2max++;
    int visits[O_MAX_A + 1];
    for (int i = 0; i < m; ++i) visits[i] = 0;
    int nc = m, len = 0;
    while (len < S) {
        int extra = (int)((double)S / (double)(log2max * nc)); /* int(num_simulations / (log2max * num_considered)) */
        if (extra < 1) extra = 1;
        for (int r = 0; r < extra; ++r) {
            for (int i = 0; i < nc && len < S; ++i) out[len++] = visits[i];
            for (int i = 0; i < nc; ++i) visits[i] += 1;
        }
        nc = nc / 2 > 2 ? nc / 2 : 2;
    }
}

static float o_max(const float* v, int n) { float m = v[0]; for (int i = 1; i < n; ++i) m = v[i] > m ? v[i] : m; return m; }
static float o_min(const float* v, int n) { float m = v[0]; for (int i = 1; i < n; ++i) m = v[i] < m ? v[i] : m; return m; }
static int o_argmax(const float* v, int n) { int bi = 0; for (int i = 1; i < n; ++i) if (v[i] > v[bi]) bi = i; return bi; }

/* jax.nn.softmax: exp(x - max) / sum */
static void o_softmax(const float* x, int n, float* p) {
    float m = o_max(x, n), e[O_MAX_A];
    for (int i = 0; i < n; ++i) e[i] = o_expf(x[i] - m);
    float s = o_tree_sum(e, n);
    for (int i = 0; i < n; ++i) p[i] = e[i] / s;
}

/* mctx _mask_invalid_actions [EXT] (same text at src/reanalyze.py:16-29) */
static void o_mask_invalid(const float* logits, const uint8_t* invalid, int n, float* out) {
    float m = o_max(logits, n);
    for (int i = 0; i < n; ++i) out[i] = invalid[i] ? O_F32_MIN : logits[i] - m;
}

/* mctx qtransforms._compute_mixed_value [EXT] */
static float o_mixed_value(float raw, const float* q, const int32_t* visits, const float* probs, int A) {
    int sumN = 0;
    float t0[O_MAX_A], t1[O_MAX_A], p2[O_MAX_A];
    for (int a = 0; a < A; ++a) {
        sumN += visits[a];
        p2[a] = probs[a] > 1.17549435e-38f ? probs[a] : 1.17549435e-38f;
        t0[a] = visits[a] > 0 ? p2[a] : 0.0f;
    }
    float sum_probs = o_tree_sum(t0, A);
    for (int a = 0; a < A; ++a) t1[a] = visits[a] > 0 ? (p2[a] * q[a]) / sum_probs : 0.0f;
    float wq = o_tree_sum(t1, A);
    return (raw + (float)sumN * wq) / ((float)sumN + 1.0f);
}

/* emctx epistemic_qtransform_completed_by_mix_value [EXT] (passed at src/selfplay.py:114-116).
 * beta == 0 (reference default, config.py:38,172) reduces to plain mctx qtransform_completed_by_mix_value. */
static void o_qtransform(const OTree* t, int b, int node, float* out) {
    const int A = t->A;
    const size_t o = ((size_t)b * t->N + node) * A;
    const float *r = t->children_rewards + o, *d = t->children_discounts + o, *cv = t->children_values + o;
    const float *rv = t->children_reward_var + o, *cvv = t->children_var + o;
    const int32_t* visits = t->children_visits + o;
    float q[O_MAX_A], qv[O_MAX_A], probs[O_MAX_A];
    for (int a = 0; a < A; ++a) {
        q[a] = r[a] + d[a] * cv[a];
        qv[a] = rv[a] + (d[a] * d[a]) * cvv[a];
    }
    o_softmax(t->children_prior_logits + o, A, probs);
    const size_t ni = (size_t)b * t->N + node;
    float value = t->raw_values[ni], value_var = t->raw_var[ni];
    if (t->use_mixed) {
        value = o_mixed_value(value, q, visits, probs, A);
        value_var = o_mixed_value(value_var, qv, visits, probs, A);
    }
    const float beta = t->beta[b];
    int maxN = 0;
    float bonus[O_MAX_A];
    for (int a = 0; a < A; ++a) {
        float c = visits[a] > 0 ? q[a] : value;
        float cvar = visits[a] > 0 ? qv[a] : value_var;
        bonus[a] = beta * sqrtf(cvar);
        out[a] = t->bonus_after_rescale ? c : c + bonus[a];
        if (visits[a] > maxN) maxN = visits[a];
    }
    if (t->rescale) {
        float mn = o_min(out, A), mx = o_max(out, A);
        float den = mx - mn;
        den = den > 1e-8f ? den : 1e-8f;
        for (int a = 0; a < A; ++a) out[a] = (out[a] - mn) / den;
    }
    if (t->bonus_after_rescale)
        for (int a = 0; a < A; ++a) out[a] = out[a] + bonus[a];
    const float scale = (t->maxvisit_init + (float)maxN) * t->value_scale;
    for (int a = 0; a < A; ++a) out[a] = scale * out[a];
}

/* mctx seq_halving.score_considered [EXT] */
static void o_score_considered(float considered_visit, const float* gumbel, const float* logits, const float* q,
                               const int32_t* visits, int A, float* out) {
    float m = o_max(logits, A);
    for (int a = 0; a < A; ++a) {
        float s = (gumbel[a] + (logits[a] - m)) + q[a];
        s = s > -1e9f ? s : -1e9f;
        out[a] = s + ((float)visits[a] == considered_visit ? 0.0f : -INFINITY);
    }
}

static int o_masked_argmax(float* v, const uint8_t* invalid, int A) {
    for (int a = 0; a < A; ++a) if (invalid[a]) v[a] = -INFINITY;
    return o_argmax(v, A);
}

/* mctx gumbel_muzero_root_action_selection [EXT] (SURVEY.md A.4 root_select) */
static int o_root_select(const OTree* t, int b) {
    const int A = t->A;
    const size_t o = ((size_t)b * t->N + 0) * A;
    float cq[O_MAX_A], sc[O_MAX_A];
    o_qtransform(t, b, 0, cq);
    int nvalid = 0, simidx = 0;
    for (int a = 0; a < A; ++a) { nvalid += 1 - (int)t->root_invalid[(size_t)b * A + a]; simidx += t->children_visits[o + a]; }
    int nc = nvalid < t->max_considered ? nvalid : t->max_considered;
    if (simidx > t->S - 1) simidx = t->S - 1;
    float cvis = (float)t->table[(size_t)nc * t->S + simidx];
    o_score_considered(cvis, t->root_gumbel + (size_t)b * A, t->children_prior_logits + o, cq, t->children_visits + o, A, sc);
    return o_masked_argmax(sc, t->root_invalid + (size_t)b * A, A);
}

/* mctx gumbel_muzero_interior_action_selection [EXT] (SURVEY.md A.4 interior_select) */
static int o_interior_select(const OTree* t, int b, int node) {
    const int A = t->A;
    const size_t o = ((size_t)b * t->N + node) * A;
    float cq[O_MAX_A], x[O_MAX_A], p[O_MAX_A];
    o_qtransform(t, b, node, cq);
    int sumN = 0;
    for (int a = 0; a < A; ++a) { x[a] = t->children_prior_logits[o + a] + cq[a]; sumN += t->children_visits[o + a]; }
    o_softmax(x, A, p);
    const float den = (float)(1 + sumN);
    for (int a = 0; a < A; ++a) x[a] = p[a] - (float)t->children_visits[o + a] / den;
    return o_argmax(x, A);
}

/* mctx search.simulate [EXT] */
static void o_simulate(const OTree* t, int b, int* parent_out, int* action_out) {
    int node = 0, depth = 0;
    for (;;) {
        int a = depth == 0 ? o_root_select(t, b) : o_interior_select(t, b, node);
        int nxt = t->children_index[((size_t)b * t->N + node) * t->A + a];
        depth += 1;
        if (nxt == -1 || depth >= t->max_depth) { *parent_out = node; *action_out = a; return; }
        node = nxt;
    }
}

/* mctx search.backward [EXT] + epistemic twin */
static void o_backward(OTree* t, int b, int leaf) {
    const int N = t->N, A = t->A;
    const size_t nb = (size_t)b * N;
    float G = t->node_values[nb + leaf], U = t->node_var[nb + leaf];
    int idx = leaf;
    while (idx != 0) {
        const int p = t->parents[nb + idx], a = t->action_from_parent[nb + idx];
        const size_t ca = (nb + p) * A + a;
        const float cnt = (float)t->node_visits[nb + p];
        const float d = t->children_discounts[ca];
        G = t->children_rewards[ca] + d * G;
        U = t->children_reward_var[ca] + (d * d) * U;
        t->node_values[nb + p] = (t->node_values[nb + p] * cnt + G) / (cnt + 1.0f);
        if (t->var_backup == 0) t->node_var[nb + p] = (t->node_var[nb + p] * cnt + U) / (cnt + 1.0f);
        else t->node_var[nb + p] = ((t->node_var[nb + p] * cnt) * cnt + U) / ((cnt + 1.0f) * (cnt + 1.0f));
        t->node_visits[nb + p] += 1;
        t->children_values[ca] = t->node_values[nb + idx];
        t->children_var[ca] = t->node_var[nb + idx];
        t->children_visits[ca] += 1;
        idx = p;
    }
}

EXPORT OTree* eazo_tree_create(const OEnv* env, const ONet* net, int B, int S) {
    OTree* t = (OTree*)calloc(1, sizeof(OTree));
    const int N = S + 1, A = env->num_actions;
    t->B = B; t->N = N; t->A = A; t->S = S; t->env = env; t->net = net; t->state_bytes = env->state_bytes;
    const size_t bn = (size_t)B * N, bna = bn * A;
#define ALLOC(p, T, n) p = (T*)calloc((n), sizeof(T))
    ALLOC(t->node_visits, int32_t, bn); ALLOC(t->parents, int32_t, bn); ALLOC(t->action_from_parent, int32_t, bn);
    ALLOC(t->raw_values, float, bn); ALLOC(t->node_values, float, bn); ALLOC(t->raw_var, float, bn); ALLOC(t->node_var, float, bn);
    ALLOC(t->children_index, int32_t, bna); ALLOC(t->children_visits, int32_t, bna);
    ALLOC(t->children_prior_logits, float, bna); ALLOC(t->children_values, float, bna); ALLOC(t->children_rewards, float, bna);
    ALLOC(t->children_discounts, float, bna); ALLOC(t->children_var, float, bna); ALLOC(t->children_reward_var, float, bna);
    ALLOC(t->beta, float, B); ALLOC(t->root_invalid, uint8_t, (size_t)B * A); ALLOC(t->root_gumbel, float, (size_t)B * A);
    ALLOC(t->embeddings, char, bn * (size_t)t->state_bytes);
    ALLOC(t->sim_parent, int32_t, B); ALLOC(t->sim_action, int32_t, B); ALLOC(t->sim_next, int32_t, B); ALLOC(t->sim_player, int32_t, B);
    ALLOC(t->obs, uint8_t, (size_t)B * env->obs_rows * env->obs_cols);
    ALLOC(t->next_states, char, (size_t)B * t->state_bytes);
    ALLOC(t->o_exploit, float, (size_t)B * A); ALLOC(t->o_explore, float, (size_t)B * A);
    ALLOC(t->o_v, float, B); ALLOC(t->o_u, float, B); ALLOC(t->o_nov, float, B);
#undef ALLOC
    return t;
}

EXPORT void eazo_tree_free(OTree* t) {
    if (!t) return;
    free(t->node_visits); free(t->parents); free(t->action_from_parent); free(t->raw_values); free(t->node_values);
    free(t->raw_var); free(t->node_var); free(t->children_index); free(t->children_visits); free(t->children_prior_logits);
    free(t->children_values); free(t->children_rewards); free(t->children_discounts); free(t->children_var);
    free(t->children_reward_var); free(t->beta); free(t->root_invalid); free(t->root_gumbel); free(t->embeddings);
    free(t->sim_parent); free(t->sim_action); free(t->sim_next); free(t->sim_player); free(t->obs); free(t->next_states);
    free(t->o_exploit); free(t->o_explore); free(t->o_v); free(t->o_u); free(t->o_nov); free(t->table);
    free(t);
}

/* field ids shared with the device library's eaz_search_tree_view (include/eaz.h) */
EXPORT void* eazo_tree_field(OTree* t, int id) {
    switch (id) {
        case 0: return t->node_visits; case 1: return t->raw_values; case 2: return t->node_values;
        case 3: return t->parents; case 4: return t->action_from_parent; case 5: return t->children_index;
        case 6: return t->children_prior_logits; case 7: return t->children_values; case 8: return t->children_visits;
        case 9: return t->children_rewards; case 10: return t->children_discounts; case 11: return t->raw_var;
        case 12: return t->node_var; case 13: return t->children_var; case 14: return t->children_reward_var;
        case 15: return t->embeddings; case 16: return t->root_gumbel; case 17: return t->beta;
        default: return NULL;
    }
}

/* mctx instantiate_tree_from_root + gumbel_muzero_policy's root masking [EXT].
 * root_logits are the RAW network logits; gumbel is already multiplied by gumbel_scale. */
/* selects the variant of the two unpinned epistemic choices for the searches that follow (default 0, 0) */
EXPORT void eazo_search_set_epistemic_variant(OTree* t, int bonus_after_rescale, int var_backup) {
    t->bonus_after_rescale = bonus_after_rescale != 0;
    t->var_backup = var_backup != 0;
}

EXPORT void eazo_search_begin(OTree* t, const float* root_logits, const float* root_value, const float* root_var,
                              const void* root_states, const float* beta, const uint8_t* invalid, const float* gumbel,
                              float discount, int two_player, int use_exploration_logits, int rescale, int use_mixed,
                              float value_scale, float maxvisit_init, int max_considered, int max_depth) {
    const int B = t->B, N = t->N, A = t->A;
    const size_t bn = (size_t)B * N, bna = bn * A;
    t->discount = discount; t->two_player = two_player; t->use_exploration_logits = use_exploration_logits;
    t->rescale = rescale; t->use_mixed = use_mixed; t->value_scale = value_scale; t->maxvisit_init = maxvisit_init;
    t->max_considered = max_considered; t->max_depth = max_depth > 0 ? max_depth : t->S;
    free(t->table);
    t->table = (int32_t*)calloc((size_t)(max_considered + 1) * t->S, sizeof(int32_t));
    for (int m = 0; m <= max_considered; ++m) eazo_considered_visits(m, t->S, t->table + (size_t)m * t->S);
    memset(t->node_visits, 0, bn * 4); memset(t->raw_values, 0, bn * 4); memset(t->node_values, 0, bn * 4);
    memset(t->raw_var, 0, bn * 4); memset(t->node_var, 0, bn * 4);
    for (size_t i = 0; i < bn; ++i) { t->parents[i] = -1; t->action_from_parent[i] = -1; }
    for (size_t i = 0; i < bna; ++i) t->children_index[i] = -1;
    memset(t->children_visits, 0, bna * 4); memset(t->children_prior_logits, 0, bna * 4); memset(t->children_values, 0, bna * 4);
    memset(t->children_rewards, 0, bna * 4); memset(t->children_discounts, 0, bna * 4); memset(t->children_var, 0, bna * 4);
    memset(t->children_reward_var, 0, bna * 4);
    memset(t->embeddings, 0, bn * (size_t)t->state_bytes);
    for (int b = 0; b < B; ++b) {
        t->beta[b] = beta[b];
        for (int a = 0; a < A; ++a) {
            t->root_invalid[(size_t)b * A + a] = invalid ? invalid[(size_t)b * A + a] : 0;
            t->root_gumbel[(size_t)b * A + a] = gumbel[(size_t)b * A + a];
        }
        const size_t nb = (size_t)b * N;
        o_mask_invalid(root_logits + (size_t)b * A, t->root_invalid + (size_t)b * A, A, t->children_prior_logits + nb * A);
        t->raw_values[nb] = t->node_values[nb] = root_value[b];
        t->raw_var[nb] = t->node_var[nb] = root_var[b];
        t->node_visits[nb] = 1;
        memcpy(t->embeddings + nb * t->state_bytes, (const char*)root_states + (size_t)b * t->state_bytes, (size_t)t->state_bytes);
    }
}

/* one simulation of mctx search.search's body_fun [EXT], with the recurrent fn of
 * src/context.py:117-155 inlined between simulate and backward. */
EXPORT void eazo_search_sim(OTree* t, int sim) {
    const int B = t->B, N = t->N, A = t->A, SB = t->state_bytes;
    const OEnv* env = t->env;
    const int D = env->obs_rows * env->obs_cols;
    OKey key = {0, 0};
    /* simulate + leaf gather + env.step (context.py:125-127) */
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < B; ++b) {
        int parent, action;
        o_simulate(t, b, &parent, &action);
        int nxt = t->children_index[((size_t)b * N + parent) * A + action];
        if (nxt == -1) nxt = sim + 1;
        t->sim_parent[b] = parent; t->sim_action[b] = action; t->sim_next[b] = nxt;
        const char* emb = t->embeddings + ((size_t)b * N + parent) * SB;
        char* ns = t->next_states + (size_t)b * SB;
        t->sim_player[b] = o_state_current_player(env, emb); /* context.py:125 */
        o_env_step(env, emb, action, key, ns);
        o_env_observe(env, ns, t->obs + (size_t)b * D);
    }
    /* forward.apply (context.py:129-131) */
    eazo_net_forward(t->net, B, t->obs, t->o_exploit, t->o_explore, t->o_v, t->o_u, t->o_nov);
    /* rest of the recurrent fn (context.py:132-153), expand (update_tree_node + topology), backward */
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const int parent = t->sim_parent[b], action = t->sim_action[b], nxt = t->sim_next[b];
        const char* ns = t->next_states + (size_t)b * SB;
        const size_t nb = (size_t)b * N;
        const float* lg = (t->use_exploration_logits ? t->o_explore : t->o_exploit) + (size_t)b * A;
        float* dst = t->children_prior_logits + (nb + nxt) * A;
        const float m = o_max(lg, A);
        for (int a = 0; a < A; ++a) dst[a] = o_state_legal(env, ns, a) ? lg[a] - m : O_F32_MIN; /* :135-137 */
        const int term = o_state_terminated(env, ns);
        const float reward = o_state_reward(env, ns, t->sim_player[b]);   /* :139 rewards[b, current_player BEFORE the step (:125)] */
        const float value = term ? 0.0f : t->o_v[b];                      /* :140 */
        const float var = term ? 0.0f : t->o_u[b];                        /* :141 */
        float disc = 1.0f * t->discount;                                  /* :142 */
        if (t->two_player) disc = disc * -1.0f;                           /* :143 */
        if (term) disc = 0.0f;                                            /* :144 */
        t->raw_values[nb + nxt] = t->node_values[nb + nxt] = value;
        t->raw_var[nb + nxt] = t->node_var[nb + nxt] = var;
        t->node_visits[nb + nxt] += 1;
        memcpy(t->embeddings + (nb + nxt) * SB, ns, (size_t)SB);
        const size_t ca = (nb + parent) * A + action;
        t->children_index[ca] = nxt;
        t->children_rewards[ca] = reward;
        t->children_reward_var[ca] = 0.0f;                                /* :149 */
        t->children_discounts[ca] = disc;
        t->parents[nb + nxt] = parent;
        t->action_from_parent[nb + nxt] = action;
        o_backward(t, b, nxt);
    }
}

/* tail of mctx gumbel_muzero_policy [EXT]: final action + action_weights */
EXPORT void eazo_search_finish(const OTree* t, int32_t* action, float* action_weights) {
    const int B = t->B, N = t->N, A = t->A;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const size_t o = ((size_t)b * N) * A;
        float cq[O_MAX_A], sc[O_MAX_A], x[O_MAX_A], m2[O_MAX_A];
        int maxN = 0;
        for (int a = 0; a < A; ++a) if (t->children_visits[o + a] > maxN) maxN = t->children_visits[o + a];
        o_qtransform(t, b, 0, cq);
        o_score_considered((float)maxN, t->root_gumbel + (size_t)b * A, t->children_prior_logits + o, cq,
                           t->children_visits + o, A, sc);
        action[b] = o_masked_argmax(sc, t->root_invalid + (size_t)b * A, A);
        for (int a = 0; a < A; ++a) x[a] = t->children_prior_logits[o + a] + cq[a];
        o_mask_invalid(x, t->root_invalid + (size_t)b * A, A, m2);
        o_softmax(m2, A, action_weights + (size_t)b * A);
    }
}

/* emctx Tree.epistemic_summary() [EXT] (consumed at src/selfplay.py:119-121,125,142) */
EXPORT void eazo_search_summary(const OTree* t, float* value, float* value_std, float* qvalues, float* qvalues_var,
                                float* visit_counts, float* visit_probs) {
    const int B = t->B, N = t->N, A = t->A;
    for (int b = 0; b < B; ++b) {
        const size_t nb = (size_t)b * N, o = nb * A;
        if (value) value[b] = t->node_values[nb];
        if (value_std) value_std[b] = sqrtf(t->node_var[nb]);
        float tot = 0.0f;
        for (int a = 0; a < A; ++a) tot += (float)t->children_visits[o + a]; /* exact: small integers */
        for (int a = 0; a < A; ++a) {
            const float d = t->children_discounts[o + a];
            if (qvalues) qvalues[(size_t)b * A + a] = t->children_rewards[o + a] + d * t->children_values[o + a];
            if (qvalues_var) qvalues_var[(size_t)b * A + a] = t->children_reward_var[o + a] + (d * d) * t->children_var[o + a];
            const float vc = (float)t->children_visits[o + a];
            if (visit_counts) visit_counts[(size_t)b * A + a] = vc;
            /* mctx Tree.summary [EXT]: visit_probs = visit_counts / max(total, 1) -- all zeros for an unsearched root */
            if (visit_probs) visit_probs[(size_t)b * A + a] = vc / (tot > 1.0f ? tot : 1.0f);
        }
    }
}

/* ===================================================================================================== *
 *  Self-play driver  (src/selfplay.py:78-149)                                                           *
 * ===================================================================================================== */

typedef struct {
    int T, S;
    float discount;
    int two_player;
    int directed_exploration;   /* config.py:38 */
    float exploration_beta;     /* config.py:69 */
    int uniform_search_policy;  /* config.py:42 */
    int sample_actions;         /* config.py:39 */
    int sample_from_improved;   /* config.py:40 */
    int rescale;                /* config.py:41 */
    float gumbel_scale;         /* emctx default 1.0 */
    int max_considered;         /* emctx default 16 */
} OSelfplayCfg;

static int o_categorical(OKey key, int b, const float* logits, int A) { /* jax.random.categorical = argmax(logits + gumbel) */
    float x[O_MAX_A];
    for (int a = 0; a < A; ++a) x[a] = logits[a] + o_gumbel(key, (uint32_t)(b * A + a));
    return o_argmax(x, A);
}

/* selfplay(): T real steps from `states` (in/out).  Outputs are [T,B,...] like SelfplayOutput (selfplay.py:17-23).
 * gumbel_in (optional, [T,B,A], unscaled) overrides the generated root noise. */
/* beta_in (optional, [B]): the per-environment beta of selfplay.py:105 given explicitly -- a SUBSET of a larger batch keeps the betas
 * beta * linspace(0, 1, B_total)[b_global] of its environments (tests/test_gpu_fullsize.py); NULL: beta * linspace(0, 1, B). */
EXPORT void eazo_selfplay_beta(const OEnv* env, const ONet* net, int B, const OSelfplayCfg* cfg, uint32_t k0, uint32_t k1,
                               void* states, const float* gumbel_in, const float* beta_in, void* out_states, float* root_value,
                               float* root_std, float* value_pred, float* ube_pred, float* qvar, int32_t* actions_out) {
    const int A = env->num_actions, D = env->obs_rows * env->obs_cols, SB = env->state_bytes, T = cfg->T;
    OTree* t = eazo_tree_create(env, net, B, cfg->S);
    uint8_t* obs = (uint8_t*)malloc((size_t)B * D);
    float *ex = (float*)malloc((size_t)B * A * 4), *xp = (float*)malloc((size_t)B * A * 4), *v = (float*)malloc((size_t)B * 4),
          *u = (float*)malloc((size_t)B * 4), *beta = (float*)malloc((size_t)B * 4), *g = (float*)malloc((size_t)B * A * 4),
          *aw = (float*)malloc((size_t)B * A * 4), *logits = (float*)malloc((size_t)B * A * 4),
          *vp = (float*)malloc((size_t)B * A * 4);
    uint8_t* invalid = (uint8_t*)calloc((size_t)B * A, 1);
    int32_t* act = (int32_t*)malloc((size_t)B * 4);
    char* nxt = (char*)malloc((size_t)B * SB);
    OKey key = {k0, k1};
    for (int step = 0; step < T; ++step) {
        OKey kt = o_fold(key, O_TAG_STEP + (uint32_t)step); /* key_seq[t], selfplay.py:147 */
        eazo_env_observe(env, B, states, obs);
        eazo_net_forward(net, B, obs, ex, xp, v, u, NULL); /* selfplay.py:89-91 */
        const float sbeta = cfg->directed_exploration ? cfg->exploration_beta : 0.0f; /* :92 */
        const float* pl = cfg->directed_exploration ? xp : ex;                        /* :93-95 */
        OKey kg = o_fold(kt, O_TAG_GUMBEL);
        for (int b = 0; b < B; ++b) {
            for (int a = 0; a < A; ++a) {
                logits[(size_t)b * A + a] = cfg->uniform_search_policy ? 1.0f : pl[(size_t)b * A + a]; /* :96-98 */
                float gn = gumbel_in ? gumbel_in[((size_t)step * B + b) * A + a] : o_gumbel(kg, (uint32_t)(b * A + a));
                g[(size_t)b * A + a] = cfg->gumbel_scale * gn;
                invalid[(size_t)b * A + a] = (uint8_t)!o_state_legal(env, (char*)states + (size_t)b * SB, a); /* :113 */
            }
            /* :105 beta * linspace(0, 1, B)[b]; linspace restated as b / (B-1) (endpoint exact) */
            beta[b] = beta_in ? (cfg->directed_exploration ? beta_in[b] : 0.0f) : sbeta * (B > 1 ? (float)b / (float)(B - 1) : 0.0f);
        }
        eazo_search_begin(t, logits, v, u, states, beta, invalid, g, cfg->discount, cfg->two_player,
                          cfg->directed_exploration, cfg->rescale, 1, 0.1f, 50.0f, cfg->max_considered, 0);
        for (int sim = 0; sim < cfg->S; ++sim) eazo_search_sim(t, sim);
        eazo_search_finish(t, act, aw);
        eazo_search_summary(t, root_value + (size_t)step * B, root_std + (size_t)step * B, NULL,
                            qvar + (size_t)step * B * A, NULL, vp);
        if (cfg->sample_actions) { /* :125,129-131 -- visit PROBABILITIES passed as logits, reproduced literally */
            OKey ks = o_fold(kt, O_TAG_SAMPLE_VISITS);
            for (int b = 0; b < B; ++b) act[b] = o_categorical(ks, b, vp + (size_t)b * A, A);
        }
        if (cfg->sample_from_improved) { /* :127,132-134 */
            OKey ks = o_fold(kt, O_TAG_SAMPLE_IMPROVED);
            for (int b = 0; b < B; ++b) act[b] = o_categorical(ks, b, aw + (size_t)b * A, A);
        }
        OKey ke = o_fold(kt, O_TAG_ENV);
        eazo_env_step(env, B, states, act, ke.k0, ke.k1, 1, nxt); /* :135 */
        memcpy(states, nxt, (size_t)B * SB);
        if (out_states) memcpy((char*)out_states + (size_t)step * B * SB, nxt, (size_t)B * SB);
        memcpy(value_pred + (size_t)step * B, v, (size_t)B * 4);
        memcpy(ube_pred + (size_t)step * B, u, (size_t)B * 4);
        if (actions_out) memcpy(actions_out + (size_t)step * B, act, (size_t)B * 4);
    }
    free(obs); free(ex); free(xp); free(v); free(u); free(beta); free(g); free(aw); free(logits); free(vp);
    free(invalid); free(act); free(nxt);
    eazo_tree_free(t);
}

EXPORT void eazo_selfplay(const OEnv* env, const ONet* net, int B, const OSelfplayCfg* cfg, uint32_t k0, uint32_t k1,
                          void* states, const float* gumbel_in, void* out_states, float* root_value,
                          float* root_std, float* value_pred, float* ube_pred, float* qvar, int32_t* actions_out) {
    eazo_selfplay_beta(env, net, B, cfg, k0, k1, states, gumbel_in, NULL, out_states, root_value, root_std, value_pred, ube_pred, qvar, actions_out);
}

/* src/selfplay.py:152-170 (uniformrandomplay): init then T uniformly random auto-reset steps */
EXPORT void eazo_random_play(const OEnv* env, int B, int T, uint32_t k0, uint32_t k1, void* states /* out: last */,
                             void* traj /* optional [T,B,state] : states BEFORE each step, as the reference returns */) {
    const int SB = env->state_bytes, A = env->num_actions;
    OKey key = {k0, k1};
    OKey ki = o_fold(key, 0xFFFFu);
    eazo_env_init(env, B, ki.k0, ki.k1, states);
    int32_t* act = (int32_t*)malloc((size_t)B * 4);
    char* nxt = (char*)malloc((size_t)B * SB);
    for (int step = 0; step < T; ++step) {
        OKey kt = o_fold(key, O_TAG_STEP + (uint32_t)step);
        OKey ka = o_fold(kt, O_TAG_RANDOM_ACTION), ke = o_fold(kt, O_TAG_ENV);
        for (int b = 0; b < B; ++b) act[b] = (int32_t)(o_bits(ka, (uint32_t)b) % (uint32_t)A);
        if (traj) memcpy((char*)traj + (size_t)step * B * SB, states, (size_t)B * SB);
        eazo_env_step(env, B, states, act, ke.k0, ke.k1, 1, nxt);
        memcpy(states, nxt, (size_t)B * SB);
    }
    free(act); free(nxt);
}

/* ===================================================================================================== *
 *  "next" rows: evaluate (src/evaluate.py:13-57) and reanalyze (src/reanalyze.py:52-131)                *
 * ===================================================================================================== */

/* evaluate(): greedy search episodes (gumbel_scale = 0, beta = exploitation_beta for every env) until all terminated or
 * counter > max_episode_length; returns the per-env sums of rewards (the reference returns their mean, :57) and the number
 * of loop iterations executed.  `states` holds the initial states (vmap(env.init), :52-54) and is advanced in place. */
EXPORT int eazo_evaluate(const OEnv* env, const ONet* net, int B, int S, float discount, int two_player, float exploitation_beta,
                         int max_episode_length, void* states, float* sum_rewards) {
    const int A = env->num_actions, D = env->obs_rows * env->obs_cols, SB = env->state_bytes;
    OTree* t = eazo_tree_create(env, net, B, S);
    uint8_t* obs = (uint8_t*)malloc((size_t)B * D);
    float *ex = (float*)malloc((size_t)B * A * 4), *xp = (float*)malloc((size_t)B * A * 4), *v = (float*)malloc((size_t)B * 4),
          *u = (float*)malloc((size_t)B * 4), *beta = (float*)malloc((size_t)B * 4), *g = (float*)calloc((size_t)B * A, 4),
          *aw = (float*)malloc((size_t)B * A * 4);
    uint8_t* invalid = (uint8_t*)calloc((size_t)B * A, 1);
    int32_t* act = (int32_t*)malloc((size_t)B * 4);
    char* nxt = (char*)malloc((size_t)B * SB);
    for (int b = 0; b < B; ++b) sum_rewards[b] = 0.0f;
    int counter = 0;
    for (;;) {
        int all_term = 1;
        for (int b = 0; b < B; ++b) all_term &= o_state_terminated(env, (char*)states + (size_t)b * SB);
        if (all_term || !(counter <= max_episode_length)) break; /* cond_fn :18-20 */
        eazo_env_observe(env, B, states, obs);
        eazo_net_forward(net, B, obs, ex, xp, v, u, NULL); /* :26-28 */
        for (int b = 0; b < B; ++b) {
            beta[b] = exploitation_beta * 1.0f; /* :34 */
            for (int a = 0; a < A; ++a) invalid[(size_t)b * A + a] = (uint8_t)!o_state_legal(env, (char*)states + (size_t)b * SB, a);
        }
        /* gumbel_scale = 0.0 (:44): the root noise is exactly zero */
        eazo_search_begin(t, ex, v, u, states, beta, invalid, g, discount, two_player, 0, 1, 1, 0.1f, 50.0f, 16, 0);
        for (int sim = 0; sim < S; ++sim) eazo_search_sim(t, sim);
        eazo_search_finish(t, act, aw);
        eazo_env_step(env, B, states, act, 0, 0, 0, nxt); /* :47 plain env.step, no auto-reset */
        for (int b = 0; b < B; ++b) /* :48-50 next_states.rewards[b, states.current_player] */
            sum_rewards[b] += o_state_reward(env, nxt + (size_t)b * SB, o_state_current_player(env, (char*)states + (size_t)b * SB));
        memcpy(states, nxt, (size_t)B * SB);
        counter += 1;
    }
    free(obs); free(ex); free(xp); free(v); free(u); free(beta); free(g); free(aw); free(invalid); free(act); free(nxt);
    eazo_tree_free(t);
    return counter;
}

typedef struct {
    int S;
    float discount;
    int two_player;
    float reanalyze_beta;                         /* config.py:75 */
    float exploration_beta;                       /* config.py:69 */
    int exploration_ube_target;                   /* config.py:33 */
    float exploration_policy_target_temperature;  /* config.py:66 */
} OReanalyzeCfg;

/* reanalyze(): targets for a batch of (state, next_state) pairs.  gumbel: explicit root noise [B,A] (unscaled; scale 1). */
EXPORT void eazo_reanalyze(const OEnv* env, const ONet* net, int B, const OReanalyzeCfg* cfg, const void* states, const void* next_states,
                           const float* gumbel, float* value_target, float* ube_target, float* exploration_policy_target,
                           float* exploitation_policy_target, int32_t* action_out) {
    const int A = env->num_actions, D = env->obs_rows * env->obs_cols, SB = env->state_bytes, S = cfg->S;
    OTree* t = eazo_tree_create(env, net, B, S);
    uint8_t* obs = (uint8_t*)malloc((size_t)B * D);
    float *ex = (float*)malloc((size_t)B * A * 4), *xp = (float*)malloc((size_t)B * A * 4), *v = (float*)malloc((size_t)B * 4),
          *u = (float*)malloc((size_t)B * 4), *nv = (float*)malloc((size_t)B * 4), *nu = (float*)malloc((size_t)B * 4),
          *beta = (float*)malloc((size_t)B * 4), *sval = (float*)malloc((size_t)B * 4), *sstd = (float*)malloc((size_t)B * 4),
          *q = (float*)malloc((size_t)B * A * 4), *qv = (float*)malloc((size_t)B * A * 4), *vc = (float*)malloc((size_t)B * A * 4);
    uint8_t* invalid = (uint8_t*)calloc((size_t)B * A, 1);
    int32_t* act = (int32_t*)malloc((size_t)B * 4);
    eazo_env_observe(env, B, states, obs);
    eazo_net_forward(net, B, obs, ex, xp, v, u, NULL); /* :64-66 */
    for (int b = 0; b < B; ++b) {
        beta[b] = 1.0f * cfg->reanalyze_beta; /* :75 */
        for (int a = 0; a < A; ++a) invalid[(size_t)b * A + a] = (uint8_t)!o_state_legal(env, (const char*)states + (size_t)b * SB, a); /* :62 */
    }
    eazo_search_begin(t, ex, v, u, states, beta, invalid, gumbel, cfg->discount, cfg->two_player, 0, 1, 1, 0.1f, 50.0f, 16, 0); /* :77-85 */
    for (int sim = 0; sim < S; ++sim) eazo_search_sim(t, sim);
    eazo_search_finish(t, act, exploitation_policy_target); /* :129 exploitation_policy_target = action_weights */
    eazo_search_summary(t, sval, sstd, q, qv, vc, NULL);
    eazo_env_observe(env, B, next_states, obs);
    eazo_net_forward(net, B, obs, xp, xp, nv, nu, NULL); /* :90-92 next_state_value */
    for (int b = 0; b < B; ++b) {
        const char* s0 = (const char*)states + (size_t)b * SB;
        const char* s1 = (const char*)next_states + (size_t)b * SB;
        const float from_tree = q[(size_t)b * A + act[b]];                                                       /* :88 */
        const float not_term1 = o_state_terminated(env, s1) ? 0.0f : 1.0f;
        const float from_td = o_state_reward(env, s1, 0) + (cfg->discount * nv[b]) * not_term1;                 /* :95-96 (single-player envs) */
        float vt = from_tree > from_td ? from_tree : from_td;                                                    /* :102 */
        float mx = qv[(size_t)b * A];
        for (int a = 1; a < A; ++a) mx = qv[(size_t)b * A + a] > mx ? qv[(size_t)b * A + a] : mx;               /* :103 */
        float ut = cfg->exploration_ube_target ? mx : qv[(size_t)b * A + act[b]];                                /* :104-106 */
        const float not_term0 = o_state_terminated(env, s0) ? 0.0f : 1.0f;
        value_target[b] = vt * not_term0;                                                                        /* :109 */
        ube_target[b] = ut * not_term0;                                                                          /* :110 */
        /* :112-122 exploration policy target: softmax(mask(complete_qs(q + b*sqrt(qvar), visits, value + b*std)) * temperature) */
        float c[O_MAX_A], m2[O_MAX_A];
        const float vfill = sval[b] + cfg->exploration_beta * sstd[b];
        for (int a = 0; a < A; ++a) {
            const float sc = q[(size_t)b * A + a] + cfg->exploration_beta * sqrtf(qv[(size_t)b * A + a]);
            c[a] = vc[(size_t)b * A + a] > 0.0f ? sc : vfill;
        }
        o_mask_invalid(c, invalid + (size_t)b * A, A, m2);
        for (int a = 0; a < A; ++a) m2[a] = m2[a] * cfg->exploration_policy_target_temperature;
        o_softmax(m2, A, exploration_policy_target + (size_t)b * A);
        if (action_out) action_out[b] = act[b];
    }
    free(obs); free(ex); free(xp); free(v); free(u); free(nv); free(nu); free(beta); free(sval); free(sstd); free(q); free(qv);
    free(vc); free(invalid); free(act);
    eazo_tree_free(t);
}

/* misc exports for tests */
EXPORT float eazo_expf(float x) { return o_expf(x); }
EXPORT float eazo_logf(float x) { return o_logf(x); }
EXPORT float eazo_tanhf(float x) { return o_tanhf(x); }
EXPORT float eazo_tree_sum(const float* v, int n) { return o_tree_sum(v, n); }
EXPORT void eazo_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* out) { o_threefry2x32(k0, k1, c0, c1, out); }
EXPORT void eazo_gumbel(uint32_t k0, uint32_t k1, int n, float* out) {
    OKey k = {k0, k1};
    for (int i = 0; i < n; ++i) out[i] = o_gumbel(k, (uint32_t)i);
}
EXPORT void eazo_fold(uint32_t k0, uint32_t k1, uint32_t tag, uint32_t* out) {
    OKey k = {k0, k1};
    OKey r = o_fold(k, tag);
    out[0] = r.k0; out[1] = r.k1;
}
EXPORT int eazo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
EXPORT void eazo_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
