// env_kernels.hpp -- batched environment kernels: init / step / auto-reset / observe / field export, and the
// leaf-gather kernel of the search (T4 "expand: gather embedding[b, parent] -> env.step -> new embedding + observation row").
//
// Leaf gather data flow (DeepSea, per tree and simulation): read the parent's 16-byte compact state, step it
// (pgx wrapper + deep_sea.py:59-81), write the child's 16-byte state into the tree's embedding slab, and emit the
// child's observation as ONE coalesced 16-byte-per-lane row store into the contiguous [B, obs_dim] uint8 inference
// batch.  The reference moves the whole pgx.State (observation included) on both sides (src/selfplay.py:104); the
// algorithmic bytes of DESIGN.md are counted on the engine's compact layout.
#pragma once
#include "envs.hpp"
#include "tree.hpp"

namespace eaz {

// ---- observation rows -------------------------------------------------------------------------------------------
// one wavefront per state; 16 bytes per lane per pass
__device__ __forceinline__ void emit_obs_row(const EnvDev& e, const char* sp, uint8_t* row, int lane) {
    const int D = e.obs_dim;
    const int hot = e.kind == EAZ_ENV_DEEPSEA ? ds_hot_index(e, ds_load(sp)) : 0;
    const bool vec = (D % 16 == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    if (vec && e.kind == EAZ_ENV_SUBLEQ) {
        // Subleq (subleq.py:679-707, encoders :26-56 / :63-98): a lane's 16 bytes span at most three row-words at >= 8 columns per word; the word
        // and its `mod W` (one integer division) are fetched when the row changes, not per byte (the generic path: 17 us per 1024 rows of
        // word_size 256)
        const eaz_subleq_state* st = reinterpret_cast<const eaz_subleq_state*>(sp);
        const int cols = e.obs_cols, W = e.W;
        for (int base = lane * 16; base < D; base += 64 * 16) {
            int r = base / cols, c = base - r * cols;
            int v = subleq_row_word(e, st, r);
            int m = pymod(v, W);
            uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint32_t byte;
                if (e.binary) byte = c == cols - 1 ? static_cast<uint32_t>(v == W) : (c < 8 ? (static_cast<uint32_t>(m & 0xFF) >> c) & 1u : 0u);  // :90-96
                else byte = static_cast<uint32_t>(c == (v == W ? W : m));                                                                     // :52
                w[i >> 2] |= byte << (8 * (i & 3));
                if (++c == cols) {
                    c = 0;
                    r += 1;
                    if (r < e.obs_rows) { v = subleq_row_word(e, st, r); m = pymod(v, W); }
                }
            }
            *reinterpret_cast<uint4*>(row + base) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else if (vec) {
        for (int base = lane * 16; base < D; base += 64 * 16) {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (e.kind == EAZ_ENV_DEEPSEA) {
                const int o = hot - base;
                if (o >= 0 && o < 16) w[o >> 2] = 1u << (8 * (o & 3));
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) w[i >> 2] |= static_cast<uint32_t>(obs_byte(e, sp, hot, base + i)) << (8 * (i & 3));
            }
            *reinterpret_cast<uint4*>(row + base) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        for (int i = lane; i < D; i += 64) row[i] = obs_byte(e, sp, hot, i);
    }
}

__global__ void __launch_bounds__(256) obs_emit_kernel(EnvDev e, StateRef src, int B, uint8_t* obs) {
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (b >= B) return;
    emit_obs_row(e, state_ptr(src, b, e.state_bytes), obs + static_cast<long>(b) * e.obs_dim, lane);
}

// ---- DeepSea ----------------------------------------------------------------------------------------------------
// mode: 0 = env.step, 1 = auto_reset(env.step, env.init) (selfplay.py:26-75), 2 = env.init
__global__ void __launch_bounds__(256) deepsea_step_kernel(EnvDev e, const char* in, const int32_t* action, int B, int mode,
                                                           char* out, char* out_copy) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    eaz_deepsea_state s;
    if (mode == 2) {
        s = ds_init();
    } else {
        s = ds_load(in + static_cast<long>(b) * sizeof(eaz_deepsea_state));
        if (mode == 1 && (s.terminated | s.truncated)) s = ds_init();
        else s = ds_step(e, s, action[b]);
    }
    ds_store(out + static_cast<long>(b) * sizeof(eaz_deepsea_state), s);
    if (out_copy) ds_store(out_copy + static_cast<long>(b) * sizeof(eaz_deepsea_state), s);
}

// leaf gather + env.step + observation emit (the UNFUSED DeepSea path; by default the tree kernel steps the environment itself and no
// observation row is written, tree.hpp TreeDev::fuse_env).  One wavefront handles kLgTrees consecutive trees: the scalar-sized loads of the
// (parent, action, child slot) triples and parent states of all of them are in flight together, then every lane writes 16 bytes of each
// tree's 1 KiB observation row (one fully coalesced 1 KiB store instruction per tree).  Round 4: with ONE tree per wavefront the launch
// topped out at 0.37 of the HBM roof however large the batch (tools/gather_sweep.py: a quarter of a million wavefronts of one store
// instruction each are bound by the rate at which wavefronts start, not by the memory system).
#ifndef EAZ_LG_TREES
#define EAZ_LG_TREES 4
#endif
constexpr int kLgTrees = EAZ_LG_TREES;
__global__ void __launch_bounds__(256) leaf_gather_deepsea_kernel(EnvDev e, TreeDev t, uint8_t* obs, float* next_reward,
                                                                  uint8_t* next_term, uint8_t* next_legal) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int b0 = w * kLgTrees;
    if (b0 >= t.B) return;
    int hot[kLgTrees];
#pragma unroll
    for (int k = 0; k < kLgTrees; ++k) {
        const bool live = b0 + k < t.B;
        const int b = live ? b0 + k : t.B - 1;
        const int parent = t.sim_parent[b], action = t.sim_action[b], nxt = t.sim_next[b];
        char* base = t.embeddings + static_cast<long>(b) * t.N * sizeof(eaz_deepsea_state);
        const eaz_deepsea_state s = ds_step(e, ds_load(base + static_cast<long>(parent) * sizeof(eaz_deepsea_state)), action);
        if (lane == 0 && live) {
            ds_store(base + static_cast<long>(nxt) * sizeof(eaz_deepsea_state), s);
            next_reward[b] = s.reward;          // rewards[b, current_player], single player (deep_sea.py:100-102)
            next_term[b] = s.terminated;
        }
        if (lane < e.A && live) next_legal[static_cast<long>(b) * e.A + lane] = 1;  // legal_action_mask all true (deep_sea.py:19)
        hot[k] = live ? ds_hot_index(e, s) : -1;
    }
    // observation rows (deep_sea.py:66-70) -> contiguous inference batch
    const int D = e.obs_dim;
#pragma unroll
    for (int k = 0; k < kLgTrees; ++k) {
        if (b0 + k >= t.B) break;
        uint8_t* row = obs + static_cast<long>(b0 + k) * D;
        if (D % 16 == 0) {
            for (int o0 = lane * 16; o0 < D; o0 += 64 * 16) {
                uint32_t wv[4] = {0u, 0u, 0u, 0u};
                const int o = hot[k] - o0;
                if (o >= 0 && o < 16) wv[o >> 2] = 1u << (8 * (o & 3));
                *reinterpret_cast<uint4*>(row + o0) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
        } else {
            for (int i = lane; i < D; i += 64) row[i] = static_cast<uint8_t>(i == hot[k]);
        }
    }
}

// reward of a SOLVED program (an unsolved one gets 0 from either function; only then have all three cases run, so `bytes_used` is the
// maximum over all of them, subleq.py:530): solved_or_not = 1, lowest_bytes = 1 / (1 + bytes_used) (one correctly rounded f32 division)
__device__ __forceinline__ float subleq_reward(const EnvDev& e, int bytes_used) {
    return e.reward_fn ? 1.0f / static_cast<float>(1 + bytes_used) : 1.0f;
}

// ---- Subleq, word sizes other than 16 ------------------------------------------------------------------------------
// One lane per environment, `envs_per_wg` (<= 64) environments per one-wavefront workgroup: the kernel's duration is the
// longest interpreter run of a wavefront, so the environments are spread over as many wavefronts / CUs as the launch has room
// for (eaz_api.hip).  The three test cases (subleq.py:511 vmap) run one after the other and stop at the first failing one:
// `solved` is the AND of the three results (:525) and only test case 0 feeds the state (`input_after` / `output_after`,
// :526-533), so a failed case makes the rest irrelevant -- exact, and two thirds of the interpreter work for the typical
// (unsolved) program.  Each run is <= 200 cycles (subleq.py:19,297-299) with an exact early exit for programs that have entered
// a cycle (subleq_simulate).  The program memory and its snapshot take 2*W bytes of LDS per
// lane.  mode as deepsea_step_kernel.  src/dst may alias (in-place real step) or be tree embedding slots (leaf gather).
// mem_copy: bytes of `memory` copied from src to dst when they differ (256, or W rounded up to 16 when the bytes above W of
// every dst slot are known to be zero: tree embeddings).
__global__ void __launch_bounds__(64) subleq_step_kernel(EnvDev e, StateRef src, StateRef dst, const int32_t* action, int B,
                                                         int mode, Key key_arg, const Key* key_ptr, char* out_copy,
                                                         float* next_reward, uint8_t* next_term, uint8_t* next_legal,
                                                         int envs_per_wg, int mem_copy) {
    const Key key = key_ptr ? *key_ptr : key_arg;  // device-resident key: lets a captured hipGraph serve every real step
    extern __shared__ __attribute__((aligned(16))) uint8_t lmem[];  // 2 * 64 * (Wp + 4) (working memory + snapshot)
    const int W = e.W;
    const int Wp = (W + 3) & ~3;  // any word_size in 16..256 (subleq.py:606): the LDS rows are whole 32-bit words, the bytes W..Wp-1 stay zero
                                  // (state bytes beyond W are zero) and no Subleq address reaches them
    const int el = threadIdx.x;
    const int b = blockIdx.x * envs_per_wg + el;
    const bool live = el < envs_per_wg && b < B;
    if (!live) return;
    const eaz_subleq_state* sp = reinterpret_cast<const eaz_subleq_state*>(state_ptr(src, b, e.state_bytes));
    eaz_subleq_state* dp = reinterpret_cast<eaz_subleq_state*>(state_ptr(dst, b, e.state_bytes));

    // ---- decide what this environment does (pgx wrapper A.1, auto_reset selfplay.py:66-71, _step subleq.py:671-677)
    bool do_init = false, noop = false, finish = false, run = false;
    int step_count = 0, task = 0, act = 0;
    {
        const bool done = mode != 2 && (sp->terminated | sp->truncated);
        do_init = mode == 2 || (mode == 1 && done);
        if (do_init) {
            const Key kb = fold(key, static_cast<uint32_t>(b));
            const Key ki = mode == 2 ? kb : fold(kb, 1u);
            const int idx = e.n_tasks > 1 ? static_cast<int>(rbits(ki, 0) % static_cast<uint32_t>(e.n_tasks)) : 0;  // subleq.py:624
            task = e.tasks[idx];
            step_count = 0;
            run = true;
        } else if (done) {
            noop = true;  // same state, zero rewards
            task = sp->task; step_count = sp->step_count;
        } else {
            task = sp->task;
            step_count = sp->step_count + 1;
            act = action[b];
            if (step_count >= W - 3 || sp->solved) finish = true;  // subleq.py:671-673
            else run = true;
        }
    }
    uint16_t tin[8], tout[8], outv[8], ex_in[8], ex_out[8], in_after[8], out_after[8];
    bool all_correct = false;
    int bytes_used = 0;
    if (run) {
        const int idx = step_count - 1;  // subleq.py:654
        all_correct = true;
        for (int c = 0; c < 3 && all_correct; ++c) {
            subleq_test_case(task, c, W, tin, tout);
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { ex_in[i] = tin[i]; ex_out[i] = tout[i]; }
            }
            // row stride Wp + 4: an odd number of 32-bit words, so that lanes reading the same program address (the common case: every
            // program starts at cursor 0) hit different LDS banks (Wp = 256 put all 64 lanes on one)
            MemLds mem{lmem + static_cast<long>(el) * (Wp + 4), lmem + static_cast<long>(64 + el) * (Wp + 4), Wp};
            if (do_init) { for (int i = 0; i < Wp; i += 4) *reinterpret_cast<uint32_t*>(mem.m + i) = 0u; }
            else {
                for (int i = 0; i < Wp; i += 4) *reinterpret_cast<uint32_t*>(mem.m + i) = *reinterpret_cast<const uint32_t*>(sp->memory + i);
                if (idx >= 0 && idx < W) mem.m[idx] = static_cast<uint8_t>(act);
            }
            int bytes_c = 0;
            const bool correct = subleq_simulate(W, mem, tin, tout, outv, bytes_c);
            bytes_used = bytes_c > bytes_used ? bytes_c : bytes_used;  // :530 results.bytes_used.max()
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { in_after[i] = tin[i]; out_after[i] = outv[i]; }
            }
            all_correct = correct;
        }
    }

    // ---- assemble the new state --------------------------------------------------------------------------------
    const bool copy_mem = reinterpret_cast<const void*>(sp) != reinterpret_cast<const void*>(dp);
    float reward;
    uint8_t terminated, solved;
    if (run) {
        solved = all_correct ? 1 : 0;                                                                 // subleq.py:525
        reward = solved ? subleq_reward(e, bytes_used) : 0.0f;                                        // :535-537 / :540-542
        terminated = do_init ? 0 : sp->terminated;  // _step never sets terminated on this branch (:661-669)
        if (do_init) {
            for (int i = 0; i < 256; i += 16) *reinterpret_cast<uint4*>(dp->memory + i) = make_uint4(0, 0, 0, 0);
        } else {
            if (copy_mem)
                for (int i = 0; i < mem_copy; i += 16) *reinterpret_cast<uint4*>(dp->memory + i) = *reinterpret_cast<const uint4*>(sp->memory + i);
            const int idx = step_count - 1;
            if (idx >= 0 && idx < W) dp->memory[idx] = static_cast<uint8_t>(act);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dp->ex_in[i] = ex_in[i];
            dp->ex_out[i] = ex_out[i];
            dp->ex_in_after[i] = in_after[i];
            dp->ex_out_after[i] = out_after[i];
        }
    } else {
        solved = sp->solved;
        reward = 0.0f;                       // terminal no-op (A.1) or subleq.py:673
        terminated = finish ? 1 : sp->terminated;
        if (copy_mem) {
            for (int i = 0; i < mem_copy; i += 16) *reinterpret_cast<uint4*>(dp->memory + i) = *reinterpret_cast<const uint4*>(sp->memory + i);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                dp->ex_in[i] = sp->ex_in[i]; dp->ex_out[i] = sp->ex_out[i];
                dp->ex_in_after[i] = sp->ex_in_after[i]; dp->ex_out_after[i] = sp->ex_out_after[i];
            }
        }
    }
    const uint8_t truncated = (do_init || !sp) ? 0 : sp->truncated;
    dp->step_count = step_count;
    dp->task = task;
    dp->reward = reward;
    dp->terminated = terminated;
    dp->truncated = truncated;
    dp->solved = solved;
    dp->pad = 0;
    (void)noop;
    if (out_copy) {
        uint4* o = reinterpret_cast<uint4*>(out_copy + static_cast<long>(b) * sizeof(eaz_subleq_state));
        const uint4* i4 = reinterpret_cast<const uint4*>(dp);
        for (int i = 0; i < static_cast<int>(sizeof(eaz_subleq_state) / 16); ++i) o[i] = i4[i];
    }
    if (next_reward) next_reward[b] = reward;
    if (next_term) next_term[b] = terminated;
    if (next_legal)
        for (int a = 0; a < e.A; ++a) next_legal[static_cast<long>(b) * e.A + a] = 1;  // all legal (subleq.py:626)
}

// ---- Subleq, word_size 16: one wavefront per environment -----------------------------------------------------------------
// A launch of the per-lane kernel above lasts as long as the slowest interpreter run in the batch, a few programs in every
// thousand run all 200 cycles, and a VALU cycle body costs ~450 clocks of a lone wavefront.  Here the machine state of ONE
// environment is wavefront-uniform, i.e. lives in scalar registers: the sixteen 4-bit words of the program as ONE 64-bit value (an
// operand fetch is a scalar shift + mask, the write-back a scalar bit-field insert), the cursor, counters and flags; the branches
// are real (no divergence: one program per wavefront) and the memory comparison of the periodicity test is one 64-bit compare.  Only
// the IO buffers are bytes of 64-bit scalars as well.
// (A first version kept the program in the lanes of a VGPR too: five v_readlane per cycle, each a VALU -> SGPR round trip in the
// cycle's dependency chain, 260 clocks per cycle.)  Results are bit-identical to subleq_simulate (same test suite); the state /
// observation stores use the 64 lanes.
__device__ __forceinline__ int sq_rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

#ifdef EAZ_FC_STAMPS
extern __device__ unsigned long long g_sq_stamps[16];
#endif
// subleq.py:297-398 `simulate` for one environment per wavefront, every piece of machine state a scalar.  smem: word l (< 16) in bits
// 4 l .. 4 l + 3.  sin / stout: input l / expected output l in byte l, W (= 16) in every byte beyond the valid ones.  Returns `correct`;
// in_after / out: input_after / output_after, same packing.  The inner loop runs the cycles that touch memory words only (both operand
// addresses <= ADDRESS_MAX, cursor in bounds): nothing of the IO state moves there, `all_eq` (:341) included, so halting is folded into
// the cycle increment `inc`.  Brent snapshots (iterations 1, 2, 4, ...; any growing schedule is exact) hold the memory in `ssnap` and
// the cursor in `s_cur`; an IO event invalidates the snapshot (the input cursor and the output only move forward, so the state before
// it cannot recur).  The IO buffers are bytes of 64-bit scalars as well (first version: lanes of VGPRs -- every IO cycle then paid
// three VALU <-> SALU round trips, and the programs the search likes best are the ones full of IO instructions).
__device__ __forceinline__ bool subleq16_wave_run(unsigned long long smem, const unsigned long long sin, const unsigned long long stout,
                                                  unsigned long long& in_after, unsigned long long& out, int& bytes_used,
                                                  int* iterations = nullptr /* interpreted cycles of this run */,
                                                  unsigned* touched = nullptr /* |= bit l for every word l fetched, read or written; kept in a VGPR */) {
    constexpr int W = 16, AMAX = 12, AIN = 13, AOUT = 14, kStop = 1000;
    constexpr unsigned long long kAllW = 0x1010101010101010ull;
    unsigned long long sout = kAllW, ssnap = 0ull;
    int cur = 0, cycles = 0, in_cur = 0, out_cursor = 0, top_cur = -3;  // top_cur: highest cursor of an executed cycle (:311)
    int s_cur = -1, s_cycle = 0, it = 0, next_snap = 1;
    int error = 0;
    int inc = sout == stout ? kStop : 1;  // :341 halts
    auto word = [&](int i) { return static_cast<int>(smem >> (4 * i)) & 15; };
    auto store = [&](int i, int v) { smem = (smem & ~(15ull << (4 * i))) | (static_cast<unsigned long long>(v) << (4 * i)); };
    // end of a cycle (:297-299 MAX_CYCLE_COUNT = 200): count it, close a period (state == snapshot => periodic with period
    // cycles - s_cycle; whole periods are skipped), snapshot
    auto cycle_end = [&]() {
        cycles += inc;
        it += 1;
        if (smem == ssnap) {  // the rare one of the three conditions first: one compare + branch on the common path
            asm volatile("");  // (keeps the compiler from folding the three tests into one mask expression evaluated every cycle)
            if (cur == s_cur && cycles < 200) {
                const int p = cycles - s_cycle;
                cycles += ((200 - cycles) / p) * p;
            }
        }
        if (it == next_snap) {
            ssnap = smem; s_cur = cur; s_cycle = cycles;
            next_snap *= 2;
        }
    };
    for (;;) {
        int a = 0, b = 0, c = 0;
        bool oob;
        for (;;) {
            oob = cur >= W - 2;  // :364-370
            if (oob) { asm volatile(""); break; }  // (the empty asm keeps the loop's two exits two branches instead of a mask expression per cycle)
            top_cur = max(top_cur, cur);
            const unsigned t = static_cast<unsigned>(smem >> (4 * cur));
            a = t & 15; b = (t >> 4) & 15; c = (t >> 8) & 15;
            if (touched) {
                // three VALU instructions on a wavefront-uniform value: the scalar port is what the interpreters of a workgroup compete for
                // (two waves per SIMD), the vector port is idle
                unsigned tv = *touched;
                asm volatile("v_lshl_or_b32 %0, 7, %1, %0\n\tv_lshl_or_b32 %0, 1, %2, %0\n\tv_lshl_or_b32 %0, 1, %3, %0" : "+v"(tv) : "s"(cur), "s"(a), "s"(b));
                *touched = tv;
            }
            if (max(a, b) > AMAX) break;
            const int v = (word(a) - word(b)) & 15;  // :322
            store(a, v);
            cur = static_cast<unsigned>(v - 1) >= 7u ? c : cur + 3;  // :329-330 jump when value == 0 or 2 * value >= 16
            cycle_end();
            if (cycles >= 200) { asm volatile(""); break; }
        }
        if (cycles >= 200) break;
        if (oob) { error = 1; break; }  // nothing else changes
        // ---- a cycle that names IN, OUT or word 15 ------------------------------------------------------------------
        // read_memory :187-232 (both reads see the pre-advance input state; the input cursor stands for the rolled array :333-338:
        // in_cur <= 8, and position 8 reads as W)
        const int in0 = in_cur < 8 ? static_cast<int>(sin >> (8 * in_cur)) & 0xFF : W;
        const bool in_ok = in0 < W;
        int ra = 0, rb = 0;
        if (a <= AMAX) ra = word(a); else if (a == AIN && in_ok) ra = in0;
        if (b <= AMAX) rb = word(b); else if (b == AIN && in_ok) rb = in0;
        const int v = (ra - rb) & 15;
        const bool uses_in = a == AIN || b == AIN;
        if (uses_in) {
            if (in_ok) { in_cur += 1; s_cur = -1; }
            else error = 1;
        }
        // write_memory :243-295
        if (a == AOUT) {
            if (out_cursor >= 8) error = 1;
            else {
                const int sh = 8 * out_cursor;
                sout = (sout & ~(0xFFull << sh)) | (static_cast<unsigned long long>(v) << sh);
                if (v != (static_cast<int>(stout >> sh) & 0xFF)) error = 1;  // :346-350
                out_cursor += 1;
                s_cur = -1;
                if (sout == stout) inc = kStop;
            }
        } else if (a <= AMAX) {
            store(a, v);
        }
        cur = static_cast<unsigned>(v - 1) >= 7u ? c : cur + 3;
        if (error) break;
        cycle_end();
        if (cycles >= 200) break;
    }
    // the rolled input (:333-338): bytes in_cur .. 7 move down, W comes in at the top
#ifdef EAZ_FC_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_sq_stamps[8] += static_cast<unsigned long long>(it); g_sq_stamps[9] += 1; if (static_cast<unsigned long long>(it) > g_sq_stamps[10]) g_sq_stamps[10] = it; }
#endif
    if (iterations) *iterations = it;
    in_after = in_cur == 0 ? sin : (in_cur >= 8 ? kAllW : ((sin >> (8 * in_cur)) | (kAllW << (64 - 8 * in_cur))));
    out = sout;
    bytes_used = top_cur + 3;
    return error == 0 && sout == stout;
}

// Novelty lookup of the observation this kernel writes (obs_out != null, XXHash variant over the IO rows = bytes 80..239): saves the
// hash_kernel launch of a simulation.  novelty == nullptr: no lookup.
struct NoveltyFuse {
    const uint8_t* bitset;
    float* novelty;
    int hash_bits;
    float nov_scale;
};

// XXHash-variant index (hashes.py:162-229) of a Subleq-16 binary observation from its 48 row-word codes (5 observation bits each: bits 0..3
// = v mod 16, bit 4 = the pad flag, subleq.py:88-98), four lanes per observation: lane i = stripe i = the contiguous quarter i of the hashed
// slice (hashes.py:211), an observation byte is 0 or 1 (bitcast float 1.0 = 0x3F800000, :207).  The slice is the four IO rows
// (subleq_hash_only_io, fully_connected.py:85-87: 32 words -> stripes of 8 words, L = 40) or, full_obs, the whole observation
// (config.py:30 False, fully_connected.py:88-89: 48 words -> stripes of 12 words, L = 60).  `words48` is 4-byte aligned.  The index is
// returned on all four lanes of the observation (lanes 4 e .. 4 e + 3 of the wavefront).
__device__ __forceinline__ uint32_t sq16_words_hash_index(const uint8_t* words48, const int i, const bool full_obs, const int hash_bits) {
    const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, SEED = 1u, ONE = 0x3F800000u * P2;
    uint32_t acc = i == 0 ? SEED + P1 + P2 : (i == 1 ? SEED + P2 : (i == 2 ? SEED : SEED - P1));  // :217-220
    const int nw = full_obs ? 12 : 8;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(words48 + (full_obs ? 12 * i : 16 + 8 * i));
#pragma unroll 1
    for (int q = 0; q < nw; q += 4) {
        const uint32_t ww = p[q >> 2];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t code = (ww >> (8 * w)) & 0xFFu;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const uint32_t t = acc + ((code >> j) & 1u ? ONE : 0u);
                acc = ((t << 13) | (t >> 19)) * P1;  // :176-181
            }
        }
    }
    const int rot = i == 0 ? 1 : (i == 1 ? 7 : (i == 2 ? 12 : 18));  // :184-187
    uint32_t h = (acc << rot) | (acc >> (32 - rot));
    h += __shfl_xor(h, 1);
    h += __shfl_xor(h, 2);
    h += full_obs ? 60u : 40u;  // :226 (stripe count L, not the byte length)
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;  // :190-197
    return h >> (32 - hash_bits);                                // :229
}

__device__ __forceinline__ void subleq16_novelty(const NoveltyFuse& nf, const int b, const uint32_t index) {
    const bool seen = nf.bitset ? ((nf.bitset[index >> 3] >> (index & 7u)) & 1u) != 0 : false;  // hashes.py:29-38
    nf.novelty[b] = (seen ? 0.0f : 1.0f) * nf.nov_scale;                                        // fully_connected.py:90
}

#ifdef EAZ_FC_STAMPS
// Diagnostic build only: shader-clock time between the stamps inside subleq16_wave_body, summed over every call of workgroup 0's
// first wave ([i] = cycles between stamp i and i + 1, [15] = calls; tools/subleq_stamps.py)
__device__ unsigned long long g_sq_stamps[16];
#define EAZ_WSTAMP(i)                                                                                  \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                     \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            if ((i) > 0) g_sq_stamps[(i)-1] += t_ - wst_prev; else g_sq_stamps[15] += 1;               \
            wst_prev = t_;                                                                             \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#define EAZ_WSTAMP_DECL unsigned long long wst_prev = 0
#else
#define EAZ_WSTAMP(i)
#define EAZ_WSTAMP_DECL
#endif

// One environment on one wavefront: state at sp -> state at dp (environment index b of the batch: key folding and the [b] outputs).
// `action`: the action of a plain step (wavefront-uniform).
__device__ __forceinline__ void subleq16_wave_body(const EnvDev& e, const char* sp, char* dp, const int b, const int lane, const int action,
                                                   const int mode, const Key key_arg, const Key* key_ptr, char* out_copy, float* next_reward,
                                                   uint8_t* next_term, uint8_t* next_legal, const int mem_copy, uint8_t* obs_out,
                                                   const NoveltyFuse& nf, uint8_t* words_out = nullptr) {
    constexpr int W = 16;
    EAZ_WSTAMP_DECL;
    EAZ_WSTAMP(0);
    const bool copy_mem = reinterpret_cast<const void*>(sp) != reinterpret_cast<const void*>(dp);
    // ---- the source state, lane-distributed: header (uniform), IO rows (lanes 0..7), memory bytes (lanes 0..15) and the
    //      16-byte memory chunks (lane l = bytes 16 l ..) for the copies
    int h_step = 0, h_task = 0, h_flags = 0;
    int s_exin = W, s_exinaf = W, s_exout = W, s_exoutaf = W, mem_raw = 0;
    uint4 chunk = make_uint4(0, 0, 0, 0);
    if (mode != 2) {
        const int4 hv = *reinterpret_cast<const int4*>(sp);
        h_step = __builtin_amdgcn_readfirstlane(hv.x);
        h_task = __builtin_amdgcn_readfirstlane(hv.y);
        h_flags = __builtin_amdgcn_readfirstlane(hv.w);
        const uint16_t* io = reinterpret_cast<const uint16_t*>(sp + 16);
        if (lane < 8) { s_exin = io[lane]; s_exinaf = io[8 + lane]; s_exout = io[16 + lane]; s_exoutaf = io[24 + lane]; }
        if (lane < W) mem_raw = reinterpret_cast<const uint8_t*>(sp + 80)[lane];
        if (16 * lane < mem_copy) chunk = *reinterpret_cast<const uint4*>(sp + 80 + 16 * lane);
    }
    EAZ_WSTAMP(1);
    const int s_term = h_flags & 0xFF, s_trunc = (h_flags >> 8) & 0xFF, s_solved = (h_flags >> 16) & 0xFF;
    // ---- decide what this environment does (pgx wrapper A.1, auto_reset selfplay.py:66-71, _step subleq.py:671-677)
    bool do_init = false, finish = false, run = false;
    int step_count = 0, task = 0, act = 0;
    {
        const bool done = mode != 2 && (s_term | s_trunc) != 0;
        do_init = mode == 2 || (mode == 1 && done);
        if (do_init) {
            const Key key = key_ptr ? *key_ptr : key_arg;
            const Key kb = fold(key, static_cast<uint32_t>(b));
            const Key ki = mode == 2 ? kb : fold(kb, 1u);
            const int idx = e.n_tasks > 1 ? static_cast<int>(rbits(ki, 0) % static_cast<uint32_t>(e.n_tasks)) : 0;  // subleq.py:624
            task = e.tasks[idx];
            step_count = 0;
            run = true;
        } else if (done) {
            task = h_task; step_count = h_step;  // same state, zero rewards
        } else {
            task = h_task;
            step_count = h_step + 1;
            act = action;
            if (step_count >= W - 3 || s_solved) finish = true;  // subleq.py:671-673
            else run = true;
        }
    }
    task = __builtin_amdgcn_readfirstlane(task);
    // ---- the new program: the action is word step_count - 1 (subleq.py:654); an initial state is all zeros
    const int idx = step_count - 1;
    if (do_init) { mem_raw = 0; chunk = make_uint4(0, 0, 0, 0); }
    const bool patch = run && !do_init && idx >= 0 && idx < W;
    if (patch) {
        const int a8 = act & 0xFF;
        if (lane == idx) mem_raw = a8;
        if (lane == 0) {
            const int sh = 8 * (idx & 3);
            uint32_t w = (idx >> 2) == 0 ? chunk.x : (idx >> 2) == 1 ? chunk.y : (idx >> 2) == 2 ? chunk.z : chunk.w;
            w = (w & ~(0xFFu << sh)) | (static_cast<uint32_t>(a8) << sh);
            if ((idx >> 2) == 0) chunk.x = w; else if ((idx >> 2) == 1) chunk.y = w; else if ((idx >> 2) == 2) chunk.z = w; else chunk.w = w;
        }
    }
    EAZ_WSTAMP(2);
    int n_exin = s_exin, n_exinaf = s_exinaf, n_exout = s_exout, n_exoutaf = s_exoutaf;
    bool all_correct = false;
    int bytes_used = 0;
    if (run) {
        all_correct = true;
        // the program as sixteen nibbles of one scalar 64-bit value: bytes & 15 of lane 0's (patched) first memory chunk
        unsigned long long smem = 0ull;
        {
            const uint32_t cw[4] = {static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(chunk.x), 0)),
                                    static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(chunk.y), 0)),
                                    static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(chunk.z), 0)),
                                    static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(chunk.w), 0))};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t w = cw[i];
                const uint32_t n4 = (w & 0xFu) | ((w >> 4) & 0xF0u) | ((w >> 8) & 0xF00u) | ((w >> 12) & 0xF000u);
                smem |= static_cast<unsigned long long>(n4) << (16 * i);
            }
        }
        for (int c = 0; c < 3 && all_correct; ++c) {  // subleq.py:511 vmap; `solved` is the AND (:525), only case 0 feeds the state
            // subleq.py:398-501 get_test_cases (lax.switch clamps the branch index): the packed vectors of the compile-time table
            const int branch = task - 1 < 0 ? 0 : (task - 1 > 5 ? 5 : task - 1);
            const unsigned long long sin = kSq16Cases.v[branch][c][0], stout = kSq16Cases.v[branch][c][1];
            unsigned long long s_after, s_out;
            int bytes_c;
            const bool correct = subleq16_wave_run(smem, sin, stout, s_after, s_out, bytes_c);
            bytes_used = max(bytes_used, bytes_c);  // :530 results.bytes_used.max()
            if (c == 0) {
                // back to the lane-distributed rows of the state (lane l < 8 = element l)
                const int sh = 8 * (lane & 7);
                n_exin = static_cast<int>(sin >> sh) & 0xFF; n_exout = static_cast<int>(stout >> sh) & 0xFF;
                n_exinaf = static_cast<int>(s_after >> sh) & 0xFF; n_exoutaf = static_cast<int>(s_out >> sh) & 0xFF;
            }
            all_correct = correct;
        }
    }
    EAZ_WSTAMP(3);
    // ---- assemble the new state --------------------------------------------------------------------------------
    float reward;
    int terminated, solved;
    if (run) {
        solved = all_correct ? 1 : 0;                  // subleq.py:525
        reward = solved ? subleq_reward(e, bytes_used) : 0.0f;  // :535-537 / :540-542
        terminated = do_init ? 0 : s_term;             // _step never sets terminated on this branch (:661-669)
    } else {
        solved = s_solved;
        reward = 0.0f;                                 // terminal no-op (A.1) or subleq.py:673
        terminated = finish ? 1 : s_term;
    }
    const int truncated = do_init ? 0 : s_trunc;
    const uint4 header = make_uint4(static_cast<uint32_t>(step_count), static_cast<uint32_t>(task), __builtin_bit_cast(uint32_t, reward),
                                    static_cast<uint32_t>(terminated) | (static_cast<uint32_t>(truncated) << 8) | (static_cast<uint32_t>(solved) << 16));
    auto emit = [&](char* base, bool io_rows, bool mem_chunks, bool mem_byte) {
        if (lane == 0) *reinterpret_cast<uint4*>(base) = header;
        if (io_rows && lane < 8) {
            uint16_t* io = reinterpret_cast<uint16_t*>(base + 16);
            io[lane] = static_cast<uint16_t>(n_exin); io[8 + lane] = static_cast<uint16_t>(n_exinaf);
            io[16 + lane] = static_cast<uint16_t>(n_exout); io[24 + lane] = static_cast<uint16_t>(n_exoutaf);
        }
        if (mem_chunks) { if (16 * lane < (do_init ? 256 : mem_copy)) *reinterpret_cast<uint4*>(base + 80 + 16 * lane) = chunk; }
        else if (mem_byte && lane == idx) reinterpret_cast<uint8_t*>(base + 80)[idx] = static_cast<uint8_t>(mem_raw);
    };
    // in place: only what changed; another slot: everything the caller asked to be copied
    emit(dp, run || copy_mem, copy_mem || do_init, patch);
    if (out_copy) emit(out_copy + static_cast<long>(b) * sizeof(eaz_subleq_state), true, true, false);
    if (lane == 0) {
        if (next_reward) next_reward[b] = reward;
        if (next_term) next_term[b] = static_cast<uint8_t>(terminated);
        if (next_legal) *reinterpret_cast<uint4*>(next_legal + static_cast<long>(b) * 16) = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);  // all legal (subleq.py:626)
    }
    EAZ_WSTAMP(4);
    if (words_out) {
        // the caller builds the observation and its hash from the 48 row-words itself (subleq16_search_kernel: straight into the forward's
        // tile, the hash on lanes that would otherwise idle): row-word w (memory | input | input_after | output | output_after,
        // subleq.py:679-707) as the five bits its observation bytes are made of -- bits 0..3 = v mod 16, bit 4 = (v == 16).  Stored by the
        // lanes that hold the values (no shuffles: the rows live on lanes 0..7, the memory words on lanes 0..15).
        auto code = [](uint32_t v) { return static_cast<uint8_t>((v & 15u) | (v == 16u ? 16u : 0u)); };
        if (lane < 16) words_out[lane] = code(static_cast<uint32_t>(mem_raw));
        if (lane < 8) {
            words_out[16 + lane] = code(static_cast<uint32_t>(n_exin) & 0xFFFFu);
            words_out[24 + lane] = code(static_cast<uint32_t>(n_exinaf) & 0xFFFFu);
            words_out[32 + lane] = code(static_cast<uint32_t>(n_exout) & 0xFFFFu);
            words_out[40 + lane] = code(static_cast<uint32_t>(n_exoutaf) & 0xFFFFu);
        }
    } else if (obs_out) {
        // binary encoding: the 48 x 5 observation (subleq.py:679-707 rows = memory | input | input_after | output | output_after;
        // :88-98 columns = 4 little-endian bits of v mod 16, then v == 16); lane w < 48 holds row-word w and stores its 5 bytes
        const int src_lane = lane & 7;
        const int r0 = __shfl(n_exin, src_lane), r1 = __shfl(n_exinaf, src_lane), r2 = __shfl(n_exout, src_lane), r3 = __shfl(n_exoutaf, src_lane);
        const int row = (lane - 16) >> 3;
        const uint32_t v = lane < 16 ? static_cast<uint32_t>(mem_raw) : (row == 0 ? r0 : row == 1 ? r1 : row == 2 ? r2 : r3) & 0xFFFFu;
        if (lane < 48) {
            uint8_t* o = obs_out + static_cast<long>(b) * 240 + 5 * lane;
            const uint32_t m = v & 15u;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = static_cast<uint8_t>((m >> j) & 1u);
            o[4] = static_cast<uint8_t>(v == 16u);
        }
        EAZ_WSTAMP(5);
        if (nf.novelty) {
            // XXHash-variant novelty of the same observation (hashes.py:162-229 over bytes 80..239 = the four IO rows, fully_connected.py:85-87):
            // hash lane i = stripe i = IO row i = 8 words x 5 bytes, every byte 0 or 1, i.e. astype(f32) bits 0 or 0x3F800000 (:207)
            const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, SEED = 1u, ONE = 0x3F800000u * P2;
            const int i = lane & 3;
            uint32_t acc = i == 0 ? SEED + P1 + P2 : (i == 1 ? SEED + P2 : (i == 2 ? SEED : SEED - P1));  // :217-220
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const uint32_t x = static_cast<uint32_t>(__shfl(static_cast<int>(v), 16 + 8 * i + w));
                const uint32_t m = x & 15u;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const bool one = j < 4 ? ((m >> j) & 1u) != 0 : x == 16u;
                    const uint32_t t = acc + (one ? ONE : 0u);
                    acc = ((t << 13) | (t >> 19)) * P1;  // :176-181
                }
            }
            const int rot = i == 0 ? 1 : (i == 1 ? 7 : (i == 2 ? 12 : 18));  // :184-187
            uint32_t h = (acc << rot) | (acc >> (32 - rot));
            h += __shfl_xor(h, 1);
            h += __shfl_xor(h, 2);
            h += 40u;  // :226 (stripe count)
            h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;  // :190-197
            const uint32_t index = h >> (32 - nf.hash_bits);             // :229
            if (lane == 0) subleq16_novelty(nf, b, index);
        }
    }
    EAZ_WSTAMP(6);
}

// Same contract as subleq_step_kernel (word_size 16 only); grid = B workgroups of one wavefront.
__global__ void __launch_bounds__(64) subleq16_wave_kernel(EnvDev e, StateRef src, StateRef dst, const int32_t* action, int B, int mode,
                                                           Key key_arg, const Key* key_ptr, char* out_copy, float* next_reward,
                                                           uint8_t* next_term, uint8_t* next_legal, int mem_copy, uint8_t* obs_out,
                                                           NoveltyFuse nf) {
    const int b = blockIdx.x;
    const int act = action ? __builtin_amdgcn_readfirstlane(action[b]) : 0;
    subleq16_wave_body(e, state_ptr(src, b, e.state_bytes), state_ptr(dst, b, e.state_bytes), b, threadIdx.x, act, mode, key_arg, key_ptr,
                       out_copy, next_reward, next_term, next_legal, mem_copy, obs_out, nf);
}

// ---- Subleq, any word size: one wavefront per environment -----------------------------------------------------------------
// The per-lane kernel above (subleq_step_kernel) spends ~1.1 k clocks per interpreted cycle at word_size 256 (a branch-free VALU body of
// ~200 dependent instructions), and a launch lasts as long as its slowest environment: a 200-cycle counting loop = 95 us.  Here the machine
// state of ONE environment is wavefront-uniform: the program memory lives in the lanes of one VGPR (lane l = bytes 4 l .. 4 l + 3, 256 bytes),
// an operand fetch is a v_readlane with a scalar index + a scalar byte extract, the write-back a compare + select on the lane id; cursor, counters, IO buffers
// (eight 16-bit words as two 64-bit scalars each) and every branch are scalar.  The periodicity test compares the memory VGPR with its
// snapshot in one v_cmp when the cursor equals the snapshot's.  subleq.py:297-398 `simulate`.
__device__ __forceinline__ bool subleqw_wave_run(const int W, const int lane, uint32_t mv, const Pack8 sin, const Pack8 stout, Pack8& in_after, Pack8& out_after,
                                                 int& bytes_used) {
    // Structure of subleq16_wave_run (above), with byte operands in the lanes of `mv` and 16-bit IO words: the inner loop runs the cycles that
    // touch memory words only; the input is not rolled but indexed by a cursor (position 8 reads as W); halting (:341) is folded into `inc`;
    // an IO event invalidates the Brent snapshot (the input cursor and the output only move forward).
    const int AMAX = W - 4, AIN = W - 3, AOUT = W - 2, kStop = 1000;
    const unsigned long long w4 = 0x0001000100010001ull * static_cast<unsigned long long>(W);
    Pack8 sout{w4, w4};
    uint32_t ssnap = 0u;
    int cur = 0, cycles = 0, in_cur = 0, out_cursor = 0, top_cur = -3;
    int s_cur = -1, s_cycle = 0, it = 0, next_snap = 1;
    int error = 0;
    int inc = pack8_eq(sout, stout) ? kStop : 1;
    auto dword = [&](int i) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mv), i)); };
    auto get16 = [](const Pack8& p, int i) { return static_cast<int>(((i < 4 ? p.lo : p.hi) >> (16 * (i & 3))) & 0xFFFFull); };
    auto diffmod = [&](int ra, int rb) {  // (ra - rb) mod W, :322: words are < W, so one conditional add; the general form only for hand-made bytes >= W
        int v = ra - rb;
        v += v < 0 ? W : 0;
        if (v < 0 || v >= W) v = pymod(ra - rb, W);
        return v;
    };
    auto cycle_end = [&]() {
        cycles += inc;
        it += 1;
        if (cur == s_cur) {
            if (cycles < 200 && __builtin_amdgcn_ballot_w64(mv != ssnap) == 0ull) {
                const int p = cycles - s_cycle;
                cycles += ((200 - cycles) / p) * p;
            }
        }
        if (it == next_snap) {
            ssnap = mv; s_cur = cur; s_cycle = cycles;
            next_snap *= 2;
        }
    };
    for (;;) {
        int a = 0, b = 0, c = 0;
        uint32_t da = 0u;
        bool oob;
        for (;;) {
            oob = cur + 2 >= W;  // :364-370
            if (oob) break;
            top_cur = top_cur > cur ? top_cur : cur;
            const int cl = cur >> 2;
            const unsigned long long two = static_cast<unsigned long long>(dword(cl)) | (static_cast<unsigned long long>(dword(cl < 63 ? cl + 1 : 63)) << 32);
            const uint32_t abc = static_cast<uint32_t>(two >> (8 * (cur & 3)));
            a = abc & 0xFF; b = (abc >> 8) & 0xFF; c = (abc >> 16) & 0xFF;
            if ((a > b ? a : b) > AMAX) break;
            da = dword(a >> 2);
            const int sha = 8 * (a & 3);
            const int v = diffmod(static_cast<int>((da >> sha) & 0xFFu), static_cast<int>((dword(b >> 2) >> (8 * (b & 3))) & 0xFFu));
            const uint32_t nd = (da & ~(0xFFu << sha)) | (static_cast<uint32_t>(v) << sha);
            mv = lane == (a >> 2) ? nd : mv;
            cur = (v == 0 || 2 * v >= W) ? c : cur + 3;  // :329-330
            cycle_end();
            if (cycles >= 200) break;
        }
        if (cycles >= 200) break;
        if (oob) { error = 1; break; }  // nothing else changes
        // ---- a cycle that names IN, OUT or a word beyond ADDRESS_MAX: read_memory :187-232 (both reads see the pre-advance input state)
        const int in0 = in_cur < 8 ? get16(sin, in_cur) : W;
        const bool in_ok = in0 < W;
        int ra = 0, rb = 0;
        if (a <= AMAX) { da = dword(a >> 2); ra = static_cast<int>((da >> (8 * (a & 3))) & 0xFFu); } else if (a == AIN && in_ok) ra = in0;
        if (b <= AMAX) rb = static_cast<int>((dword(b >> 2) >> (8 * (b & 3))) & 0xFFu); else if (b == AIN && in_ok) rb = in0;
        const int v = diffmod(ra, rb);
        if (a == AIN || b == AIN) {
            if (in_ok) { in_cur += 1; s_cur = -1; }
            else error = 1;
        }
        // write_memory :243-295
        if (a == AOUT) {
            if (out_cursor >= 8) error = 1;
            else {
                const int sh = 16 * (out_cursor & 3);
                const unsigned long long clr = ~(0xFFFFull << sh), ins = static_cast<unsigned long long>(v) << sh;
                if (out_cursor < 4) sout.lo = (sout.lo & clr) | ins; else sout.hi = (sout.hi & clr) | ins;
                if (v != get16(stout, out_cursor)) error = 1;  // :346-350
                out_cursor += 1;
                s_cur = -1;
                if (pack8_eq(sout, stout)) inc = kStop;
            }
        } else if (a <= AMAX) {
            const int sha = 8 * (a & 3);
            const uint32_t nd = (da & ~(0xFFu << sha)) | (static_cast<uint32_t>(v) << sha);
            mv = lane == (a >> 2) ? nd : mv;
        }
        cur = (v == 0 || 2 * v >= W) ? c : cur + 3;
        if (error) break;
        cycle_end();
        if (cycles >= 200) break;
    }
    // the rolled input (:333-338): words in_cur .. 7 move down, W comes in at the top
    Pack8 ia = sin;
    for (int k = 0; k < in_cur; ++k) {
        ia.lo = (ia.lo >> 16) | (ia.hi << 48);
        ia.hi = (ia.hi >> 16) | (static_cast<unsigned long long>(W) << 48);
    }
    in_after = ia;
    out_after = sout;
    bytes_used = top_cur + 3;
    return error == 0 && pack8_eq(sout, stout);
}

// env.step of environment `b` by one wavefront: state blob `sp` -> `dp` (mode / keys / outputs as subleq_step_kernel; `act_in` is read when a
// step is taken).  obs_out: the binary observation row; nf.novelty: its novelty where the hashed slice is the four IO rows
// (fully_connected.py:85-87: lane q = stripe q = IO row q, 8 words x obs_cols bytes, hashes.py:162-229) -- the caller checks both.
__device__ __forceinline__ void subleq_wave_body(const EnvDev& e, const char* sp, char* dp, const int b, const int lane, const int act_in, const int mode,
                                                 const Key key_arg, const Key* key_ptr, char* out_copy, float* next_reward, uint8_t* next_term,
                                                 uint8_t* next_legal, uint8_t* obs_out, const NoveltyFuse& nf) {
    const int W = e.W;
    const eaz_subleq_state* ss = reinterpret_cast<const eaz_subleq_state*>(sp);
    int h_step = 0, h_task = 0, h_flags = 0;
    uint32_t md = 0u;
    if (mode != 2) {
        const int4 hv = *reinterpret_cast<const int4*>(sp);
        h_step = __builtin_amdgcn_readfirstlane(hv.x);
        h_task = __builtin_amdgcn_readfirstlane(hv.y);
        h_flags = __builtin_amdgcn_readfirstlane(hv.w);
        md = reinterpret_cast<const uint32_t*>(sp + 80)[lane];
    }
    const int s_term = h_flags & 0xFF, s_trunc = (h_flags >> 8) & 0xFF, s_solved = (h_flags >> 16) & 0xFF;
    // ---- decide what this environment does (pgx wrapper A.1, auto_reset selfplay.py:66-71, _step subleq.py:671-677)
    bool do_init = false, finish = false, run = false;
    int step_count = 0, task = 0, act = 0;
    {
        const bool done = mode != 2 && (s_term | s_trunc) != 0;
        do_init = mode == 2 || (mode == 1 && done);
        if (do_init) {
            const Key key = key_ptr ? *key_ptr : key_arg;
            const Key kb = fold(key, static_cast<uint32_t>(b));
            const Key ki = mode == 2 ? kb : fold(kb, 1u);
            const int idx = e.n_tasks > 1 ? static_cast<int>(rbits(ki, 0) % static_cast<uint32_t>(e.n_tasks)) : 0;  // subleq.py:624
            task = e.tasks[idx];
            step_count = 0;
            run = true;
            md = 0u;
        } else if (done) {
            task = h_task; step_count = h_step;  // same state, zero rewards
        } else {
            task = h_task;
            step_count = h_step + 1;
            act = act_in;
            if (step_count >= W - 3 || s_solved) finish = true;  // subleq.py:671-673
            else run = true;
        }
    }
    task = __builtin_amdgcn_readfirstlane(task);
    Pack8 ex_in{0ull, 0ull}, ex_out{0ull, 0ull}, in_after{0ull, 0ull}, out_after{0ull, 0ull};
    bool all_correct = false;
    int bytes_used = 0;
    if (run) {
        const int idx = step_count - 1;  // the action is word step_count - 1 (subleq.py:654)
        if (!do_init && idx >= 0 && idx < W && lane == (idx >> 2)) md = (md & ~(0xFFu << (8 * (idx & 3)))) | ((static_cast<uint32_t>(act) & 0xFFu) << (8 * (idx & 3)));
        all_correct = true;
        for (int c = 0; c < 3 && all_correct; ++c) {  // subleq.py:511 vmap; `solved` is the AND (:525), only case 0 feeds the state
            Pack8 pin, pout;
            subleq_test_case_packed(task, c, W, pin, pout);
            Pack8 ia, oa;
            int bytes_c = 0;
            const bool correct = subleqw_wave_run(W, lane, md, pin, pout, ia, oa, bytes_c);
            bytes_used = bytes_c > bytes_used ? bytes_c : bytes_used;  // :530
            if (c == 0) { ex_in = pin; ex_out = pout; in_after = ia; out_after = oa; }
            all_correct = correct;
        }
    }
    // ---- assemble the new state --------------------------------------------------------------------------------
    const bool copy_mem = reinterpret_cast<const void*>(sp) != reinterpret_cast<const void*>(dp);
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
    const uint4 header = make_uint4(static_cast<uint32_t>(step_count), static_cast<uint32_t>(task), __builtin_bit_cast(uint32_t, reward),
                                    static_cast<uint32_t>(terminated) | (static_cast<uint32_t>(truncated) << 8) | (static_cast<uint32_t>(solved) << 16));
    uint4 row0, row1, row2, row3;  // (four named registers sets: an indexed array of them lived in scratch memory)
    if (run) {
        auto r4 = [](const Pack8& p) { return make_uint4(static_cast<uint32_t>(p.lo), static_cast<uint32_t>(p.lo >> 32), static_cast<uint32_t>(p.hi), static_cast<uint32_t>(p.hi >> 32)); };
        row0 = r4(ex_in); row1 = r4(in_after); row2 = r4(ex_out); row3 = r4(out_after);
    } else {
        row0 = *reinterpret_cast<const uint4*>(sp + 16); row1 = *reinterpret_cast<const uint4*>(sp + 32);
        row2 = *reinterpret_cast<const uint4*>(sp + 48); row3 = *reinterpret_cast<const uint4*>(sp + 64);
    }
    (void)ss;
    auto emit = [&](char* base, bool everything) {
        if (lane == 0) {
            *reinterpret_cast<uint4*>(base) = header;
            if (everything || run) {
                *reinterpret_cast<uint4*>(base + 16) = row0; *reinterpret_cast<uint4*>(base + 32) = row1;
                *reinterpret_cast<uint4*>(base + 48) = row2; *reinterpret_cast<uint4*>(base + 64) = row3;
            }
        }
        if (everything || run) reinterpret_cast<uint32_t*>(base + 80)[lane] = md;
    };
    emit(dp, copy_mem);
    if (out_copy) emit(out_copy + static_cast<long>(b) * sizeof(eaz_subleq_state), true);
    if (lane == 0) {
        if (next_reward) next_reward[b] = reward;
        if (next_term) next_term[b] = static_cast<uint8_t>(terminated);
    }
    if (next_legal)
        for (int a = lane; a < e.A; a += 64) next_legal[static_cast<long>(b) * e.A + a] = 1;  // all legal (subleq.py:626)
    if (obs_out) {
        // binary encoding only (the caller checks): the observation row of the new state (subleq.py:679-707 rows = memory | example_input |
        // input_after | example_output | output_after; :88-98 columns = little-endian bits of v mod W, then v == W) -- no obs_emit launch
        uint8_t* o = obs_out + static_cast<long>(b) * e.obs_dim;
        const int cols = e.obs_cols;
        for (int i = lane; i < ((W + 32 + 63) & ~63); i += 64) {
            const uint32_t mdw = static_cast<uint32_t>(__shfl(static_cast<int>(md), (i < W ? i : 0) >> 2));
            int v;
            if (i < W) v = static_cast<int>((mdw >> (8 * (i & 3))) & 0xFFu);
            else {
                const int r = (i - W) >> 3, k = (i - W) & 7;
                const uint4 rw = r == 0 ? row0 : (r == 1 ? row1 : (r == 2 ? row2 : row3));
                const uint32_t pair = (k >> 1) == 0 ? rw.x : ((k >> 1) == 1 ? rw.y : ((k >> 1) == 2 ? rw.z : rw.w));
                v = static_cast<int>((pair >> (16 * (k & 1))) & 0xFFFFu);
            }
            if (i < W + 32) {
                const unsigned m = static_cast<unsigned>(pymod(v, W)) & 0xFFu;  // :90
                uint8_t* ow = o + i * cols;
                for (int j = 0; j < cols - 1; ++j) ow[j] = j < 8 ? static_cast<uint8_t>((m >> j) & 1u) : static_cast<uint8_t>(0);  // :91
                ow[cols - 1] = static_cast<uint8_t>(v == W);                                                                       // :96
            }
        }
    }
    if (nf.novelty) {
        // XXHash variant over the IO rows (hashes.py:162-229): lane q < 4 = stripe q = row q, every observation byte is 0 or 1 (bitcast 1.0f)
        const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, SEED = 1u, ONE = 0x3F800000u * P2;
        const int q = lane & 3, cols = e.obs_cols;
        uint32_t acc = q == 0 ? SEED + P1 + P2 : (q == 1 ? SEED + P2 : (q == 2 ? SEED : SEED - P1));  // :217-220
        const uint4 rw = q == 0 ? row0 : (q == 1 ? row1 : (q == 2 ? row2 : row3));
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
            const uint32_t pair = (k >> 1) == 0 ? rw.x : ((k >> 1) == 1 ? rw.y : ((k >> 1) == 2 ? rw.z : rw.w));
            const int v = static_cast<int>((pair >> (16 * (k & 1))) & 0xFFFFu);
            const unsigned m = static_cast<unsigned>(pymod(v, W)) & 0xFFu;  // :90
#pragma unroll 1
            for (int j = 0; j < cols - 1; ++j) {
                const uint32_t t = acc + ((j < 8 && ((m >> j) & 1u)) ? ONE : 0u);
                acc = ((t << 13) | (t >> 19)) * P1;  // :176-181
            }
            const uint32_t t = acc + (v == W ? ONE : 0u);  // :96
            acc = ((t << 13) | (t >> 19)) * P1;
        }
        const int rot = q == 0 ? 1 : (q == 1 ? 7 : (q == 2 ? 12 : 18));  // :184-187
        uint32_t h = (acc << rot) | (acc >> (32 - rot));
        h += __shfl_xor(h, 1);
        h += __shfl_xor(h, 2);
        h += static_cast<uint32_t>(8 * cols);  // :226 (stripe count)
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;  // :190-197
        const uint32_t index = nf.hash_bits >= 32 ? h : h >> (32 - nf.hash_bits);  // :229
        if (lane == 0) subleq16_novelty(nf, b, index);
    }
}

// Same contract as subleq_step_kernel; grid = B workgroups of one wavefront.  Lane l carries bytes 4 l .. 4 l + 3 of the state's 256-byte
// memory field (all of it: the bytes beyond word_size are zero and travel with the copy), lane 0 the header and the four IO rows.
__global__ void __launch_bounds__(64) subleq_wave_kernel(EnvDev e, StateRef src, StateRef dst, const int32_t* action, int B, int mode, Key key_arg,
                                                         const Key* key_ptr, char* out_copy, float* next_reward, uint8_t* next_term,
                                                         uint8_t* next_legal, uint8_t* obs_out, NoveltyFuse nf) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int act = (action && mode != 2) ? __builtin_amdgcn_readfirstlane(action[b]) : 0;
    subleq_wave_body(e, state_ptr(src, b, e.state_bytes), state_ptr(dst, b, e.state_bytes), b, lane, act, mode, key_arg, key_ptr, out_copy, next_reward,
                     next_term, next_legal, obs_out, nf);
}

// ---- connect_four -------------------------------------------------------------------------------------------------
// One thread per environment, the game on bitboards (envs.hpp c4_bits_*: no indexed local array, no scratch).  mode as deepsea_step_kernel.
// mover_reward[b] = rewards[b, current_player BEFORE the step] (context.py:125,139 / evaluate.py:48-50).
__global__ void __launch_bounds__(256) connect4_step_kernel(EnvDev e, StateRef src, StateRef dst, const int32_t* action, int B, int mode,
                                                            Key key_arg, const Key* key_ptr, float* mover_reward, uint8_t* next_term,
                                                            uint8_t* next_legal) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Key key = key_ptr ? *key_ptr : key_arg;
    C4Bits s;
    int mover = 0;
    if (mode == 2) {
        c4_bits_init(s, fold(key, static_cast<uint32_t>(b)));
    } else {
        const uint4* sp = reinterpret_cast<const uint4*>(state_ptr(src, b, e.state_bytes));
        uint4 v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = sp[i];
        c4_bits_load(s, v);
        mover = s.current_player;
        if (mode == 1 && (s.terminated | s.truncated)) c4_bits_init(s, fold(fold(key, static_cast<uint32_t>(b)), 1u));
        else c4_bits_step(s, action[b]);
    }
    uint4* dp = reinterpret_cast<uint4*>(state_ptr(dst, b, e.state_bytes));
#pragma unroll
    for (int i = 0; i < 5; ++i) dp[i] = make_uint4(c4_bits_word(s, 4 * i), c4_bits_word(s, 4 * i + 1), c4_bits_word(s, 4 * i + 2), c4_bits_word(s, 4 * i + 3));
    if (mover_reward) mover_reward[b] = mover == 0 ? s.reward0 : s.reward1;
    if (next_term) next_term[b] = static_cast<uint8_t>(s.terminated);
    if (next_legal)
        for (int a = 0; a < 7; ++a) next_legal[static_cast<long>(b) * 7 + a] = c4_bits_legal(s, a) ? 1 : 0;
}

// The leaf step of a connect_four simulation in ONE launch (context.py:125-131 up to the forward): gather the parent state, env.step of the
// selected edge, child state into its embedding slot, the [6, 7, 2] observation row of the inference batch, mover reward / terminated / legal
// mask, and the novelty lookup of the new observation (XXHash variant over the whole row: four lanes per environment, lane i = the contiguous
// quarter i = 21 observation bytes, hashes.py:162-229) -- what connect4_step_kernel + obs_emit_kernel + hash_kernel (on a side stream,
// forked and joined per simulation) did in three.  Four lanes per environment, each computes the step (a few dozen integer operations on
// two 64-bit boards), the stores are shared out.  novelty == nullptr (SimHash networks): no lookup, the caller runs its hash kernel.
__global__ void __launch_bounds__(64) connect4_leaf_kernel(EnvDev e, StateRef src, StateRef dst, const int32_t* action, int B, float* mover_reward,
                                                           uint8_t* next_term, uint8_t* next_legal, uint8_t* obs, NoveltyFuse nf) {
    const int gt = blockIdx.x * 64 + threadIdx.x;
    const int b = gt >> 2, q = gt & 3;
    const bool live = b < B;
    const int bb = live ? b : B - 1;
    C4Bits s;
    {
        const uint4* sp = reinterpret_cast<const uint4*>(state_ptr(src, bb, e.state_bytes));
        uint4 v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = sp[i];
        c4_bits_load(s, v);
    }
    const int mover = s.current_player;
    c4_bits_step(s, action[bb]);
    // child state: lane q writes 16-byte words q and q + 4 of the blob's five
    uint4* dp = reinterpret_cast<uint4*>(state_ptr(dst, bb, e.state_bytes));
    if (live) {
        dp[q] = make_uint4(c4_bits_word(s, 4 * q), c4_bits_word(s, 4 * q + 1), c4_bits_word(s, 4 * q + 2), c4_bits_word(s, 4 * q + 3));
        if (q == 0) dp[4] = make_uint4(c4_bits_word(s, 16), 0u, 0u, 0u);
    }
    // observation row (84 bytes = 21 words): lane q writes words 6 q .. 6 q + 5 (lane 3: three)
    if (live) {
        uint32_t* orow = reinterpret_cast<uint32_t*>(obs + static_cast<long>(b) * 84);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int w = 6 * q + k;
            if (w < 21) {
                uint32_t word = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) word |= c4_bits_obs(s, 4 * w + j) << (8 * j);
                orow[w] = word;
            }
        }
    }
    if (nf.novelty) {
        const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, SEED = 1u, ONE = 0x3F800000u * P2;
        uint32_t acc = q == 0 ? SEED + P1 + P2 : (q == 1 ? SEED + P2 : (q == 2 ? SEED : SEED - P1));  // :217-220
#pragma unroll
        for (int l = 0; l < 21; ++l) {
            const uint32_t t = acc + (c4_bits_obs(s, 21 * q + l) ? ONE : 0u);
            acc = ((t << 13) | (t >> 19)) * P1;  // :176-181
        }
        const int rot = q == 0 ? 1 : (q == 1 ? 7 : (q == 2 ? 12 : 18));  // :184-187
        uint32_t h = (acc << rot) | (acc >> (32 - rot));
        h += __shfl_xor(h, 1);
        h += __shfl_xor(h, 2);
        h += 21u;  // :226 (stripe count)
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;  // :190-197
        const uint32_t index = nf.hash_bits >= 32 ? h : h >> (32 - nf.hash_bits);  // :229
        if (q == 0 && live) subleq16_novelty(nf, b, index);
    }
    if (live && q == 1) {
        if (mover_reward) mover_reward[b] = mover == 0 ? s.reward0 : s.reward1;
        if (next_term) next_term[b] = static_cast<uint8_t>(s.terminated);
    }
    if (live && q == 2 && next_legal)
        for (int a = 0; a < 7; ++a) next_legal[static_cast<long>(b) * 7 + a] = c4_bits_legal(s, a) ? 1 : 0;
}

// ---- state field export (pgx.State-visible fields) -----------------------------------------------------------------
__global__ void __launch_bounds__(256) env_fields_kernel(EnvDev e, const char* states, int B, uint8_t* terminated,
                                                         uint8_t* truncated, float* rewards, int32_t* step_count, uint8_t* legal,
                                                         int32_t* current_player) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const char* sp = states + static_cast<long>(b) * e.state_bytes;
    const StateHeader* h = reinterpret_cast<const StateHeader*>(sp);
    if (terminated) terminated[b] = h->terminated;
    if (truncated) truncated[b] = h->truncated;
    if (step_count) step_count[b] = h->step_count;
    if (e.kind == EAZ_ENV_CONNECT4) {
        const eaz_connect4_state* c = reinterpret_cast<const eaz_connect4_state*>(sp);
        if (rewards) { rewards[2 * b] = c->reward0; rewards[2 * b + 1] = c->reward1; }
        if (current_player) current_player[b] = c->current_player;
        if (legal)
            for (int a = 0; a < 7; ++a) legal[static_cast<long>(b) * 7 + a] = c4_legal(*c, a) ? 1 : 0;
    } else {
        if (rewards) rewards[b] = h->reward0;
        if (current_player) current_player[b] = 0;
        if (legal)
            for (int a = 0; a < e.A; ++a) legal[static_cast<long>(b) * e.A + a] = 1;
    }
}

}  // namespace eaz
