// envs.hpp -- device-side environments: pgx.Env wrapper semantics + DeepSea + Subleq.
//
// Reference: pgx.Env.step/init/observe [EXT] (called at src/context.py:127, src/selfplay.py:135,166),
//            src/envs/deep_sea.py:54-85, src/envs/subleq.py:26-98,156-532,623-707, auto_reset src/selfplay.py:26-75.
// States are the compact blobs of include/eaz.h; the observation is regenerated from them on demand.
#pragma once
#include "../../include/eaz.h"
#include "rng.hpp"

namespace eaz {

struct EnvDev {
    int kind;
    int A;
    int obs_rows, obs_cols, obs_dim;
    int state_bytes;
    // DeepSea
    int G;
    const uint8_t* action_map;  // device [G*G]
    // Subleq
    int W;
    int n_tasks;
    int tasks[16];
    int binary;
    int reward_fn;  // 0 = solved_or_not (subleq.py:535-537), 1 = lowest_bytes (:540-542)
};

// A batch of states addressed as base + (b * nodes_per_tree + idx[b]) * state_bytes  (idx == nullptr -> 0).
struct StateRef {
    char* base;
    const int32_t* idx;
    int nodes_per_tree;
};
__device__ __forceinline__ char* state_ptr(const StateRef& r, int b, int state_bytes) {
    const long n = r.idx ? r.idx[b] : 0;
    return r.base + (static_cast<long>(b) * r.nodes_per_tree + n) * state_bytes;
}

// ---------------------------------------------------------------------------------------------------------------
// DeepSea
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ eaz_deepsea_state ds_load(const char* p) {
    const int4 v = *reinterpret_cast<const int4*>(p);
    eaz_deepsea_state s;
    s.step_count = v.x;
    s.pos = v.y;
    s.reward = __builtin_bit_cast(float, v.z);
    s.terminated = static_cast<uint8_t>(v.w & 0xFF);
    s.truncated = static_cast<uint8_t>((v.w >> 8) & 0xFF);
    s.pad[0] = s.pad[1] = 0;
    return s;
}
__device__ __forceinline__ void ds_store(char* p, const eaz_deepsea_state& s) {
    int4 v;
    v.x = s.step_count;
    v.y = s.pos;
    v.z = __builtin_bit_cast(int, s.reward);
    v.w = static_cast<int>(s.terminated) | (static_cast<int>(s.truncated) << 8);
    *reinterpret_cast<int4*>(p) = v;
}

// deep_sea.py:54-57
__device__ __forceinline__ eaz_deepsea_state ds_init() {
    eaz_deepsea_state s;
    s.step_count = 0; s.pos = 0; s.reward = 0.0f; s.terminated = 0; s.truncated = 0; s.pad[0] = s.pad[1] = 0;
    return s;
}

// pgx.Env.step wrapper (A.1) around deep_sea.py:59-81.  All actions are legal (deep_sea.py:19).
__device__ __forceinline__ eaz_deepsea_state ds_step(const EnvDev& e, eaz_deepsea_state s, int action) {
    if (s.terminated | s.truncated) {  // terminal no-op with zero rewards
        s.reward = 0.0f;
        return s;
    }
    const int G = e.G;
    s.step_count += 1;
    int r = s.step_count - 1;
    r = r < 0 ? 0 : (r > G - 1 ? G - 1 : r);
    // (the column of a hand-made blob may be anything: clamped like the row -- jnp clamps an out-of-range index, and this read stays inside the map)
    const int pc = s.pos < 0 ? 0 : (s.pos > G - 1 ? G - 1 : s.pos);
    const int flip = e.action_map[r * G + pc] != 0;             // :62
    const int shift = ((action == 0) ^ flip) ? -1 : 1;          // :63
    int pos = s.pos + shift;
    pos = pos < 0 ? 0 : (pos > G - 1 ? G - 1 : pos);            // :64
    const int term = s.step_count >= G - 1;                     // :72
    s.pos = pos;
    s.terminated = static_cast<uint8_t>(term);
    s.reward = (term && pos == G - 1) ? 1.0f : 0.0f;            // :74-78
    return s;
}

// flat index of the single True cell, deep_sea.py:66-70
// (-1 = no cell: a hand-made blob with a negative step count or a column off the grid observes all zeros -- an index outside the array
// addresses nothing; states the environment produced are always on the grid)
__device__ __forceinline__ int ds_hot_index(const EnvDev& e, const eaz_deepsea_state& s) {
    const int r = s.step_count < e.G - 1 ? s.step_count : e.G - 1;
    return (r >= 0 && s.pos >= 0 && s.pos < e.G) ? r * e.G + s.pos : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// Subleq
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int pymod(int x, int m) {
    const int r = x % m;
    return r < 0 ? r + m : r;
}

// subleq.py:398-501 get_test_cases, the tables themselves: test case `c` of branch `branch` (= task - 1 clamped to 0..5: lax.switch clamps the
// index, so task ids >= 6 all select the multiplication table) as raw integers, before the modulo and the padding (:101-107,:402-404).  ONE
// statement of the tables for the host (eaz_subleq_test_vectors, held against the oracle's generator by tests/test_abi_and_host.py), the
// word_size-16 interpreter's packed table (kSq16Cases) and the other word sizes' table (kSqRaw).
struct SqRawCase {
    int in[8], out[8];
    int nin, nout;
};
constexpr SqRawCase sq_raw_case(int branch, int c) {
    SqRawCase r{{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, 0, 0};
    if (branch == 0) {  // NEGATION_POSITIVE :406-411
        const int t1[8] = {5, 4, 4, 5, 1, 2, 3, 1}, t2[8] = {1, 1, 6, 2, 4, 4, 5, 3};
        if (c == 0) { r.nin = 7; for (int i = 0; i < 7; ++i) r.in[i] = i + 1; }
        else { r.nin = 8; for (int i = 0; i < 8; ++i) r.in[i] = c == 1 ? t1[i] : t2[i]; }
        r.nout = r.nin;
        for (int i = 0; i < 8; ++i) r.out[i] = -r.in[i];
    } else if (branch == 1 || branch == 2) {  // NEGATION :413-418 / IDENTITY :420-421
        const int t2[5] = {0, -1, 2, -3, 4};
        if (c == 0) { r.nin = 8; for (int i = 0; i < 8; ++i) r.in[i] = -4 + i; }
        else if (c == 1) { r.nin = 7; for (int i = 0; i < 7; ++i) r.in[i] = i + 1; }
        else { r.nin = 5; for (int i = 0; i < 5; ++i) r.in[i] = t2[i]; }
        r.nout = r.nin;
        for (int i = 0; i < 8; ++i) r.out[i] = branch == 1 ? -r.in[i] : r.in[i];
    } else if (branch == 3 || branch == 4) {  // SUBTRACTION :423-432 / ADDITION :434-443
        const int a0[6] = {1, 1, 5, 4, 0, -3}, a1[6] = {2, 3, 0, 0, -1, -2}, a2[8] = {1, 2, 3, 4, 4, 3, 2, 1};
        const int s0[3] = {0, 1, 3}, s1[3] = {-1, 0, 1}, s2[4] = {-1, -1, 1, 1};
        const int p0[3] = {2, 9, -3}, p1[3] = {5, 0, -3}, p2[4] = {3, 7, 7, 3};
        if (c == 0) { r.nin = 6; r.nout = 3; for (int i = 0; i < 6; ++i) r.in[i] = a0[i]; for (int i = 0; i < 3; ++i) r.out[i] = branch == 3 ? s0[i] : p0[i]; }
        else if (c == 1) { r.nin = 6; r.nout = 3; for (int i = 0; i < 6; ++i) r.in[i] = a1[i]; for (int i = 0; i < 3; ++i) r.out[i] = branch == 3 ? s1[i] : p1[i]; }
        else { r.nin = 8; r.nout = 4; for (int i = 0; i < 8; ++i) r.in[i] = a2[i]; for (int i = 0; i < 4; ++i) r.out[i] = branch == 3 ? s2[i] : p2[i]; }
    } else {  // MULTIPLICATION :445-454
        const int m0[6] = {1, 2, 2, 3, -7, 4}, m1[6] = {-1, -5, 5, 2, 0, 1}, m2[6] = {0, 0, 1, 1, -3, 3};
        const int o0[3] = {2, 6, -28}, o1[3] = {5, 10, 0}, o2[3] = {0, 1, -9};
        r.nin = 6; r.nout = 3;
        for (int i = 0; i < 6; ++i) r.in[i] = c == 0 ? m0[i] : (c == 1 ? m1[i] : m2[i]);
        for (int i = 0; i < 3; ++i) r.out[i] = c == 0 ? o0[i] : (c == 1 ? o1[i] : o2[i]);
    }
    return r;
}
__host__ __device__ constexpr int sq_task_branch(int task) { return task - 1 < 0 ? 0 : (task - 1 > 5 ? 5 : task - 1); }

// the raw tables packed for the device (any word size): eight signed bytes per row in one 64-bit word -- a case is read with scalar loads
// and its elements come out by shifts; an indexed local array (SqRawCase by value) would live in scratch memory
struct SqRawPacked {
    unsigned long long in, out;
    int nin, nout;
};
struct SqRawTable {
    SqRawPacked v[6][3];
};
constexpr SqRawTable sq_make_raw() {
    SqRawTable t{};
    for (int branch = 0; branch < 6; ++branch)
        for (int c = 0; c < 3; ++c) {
            const SqRawCase r = sq_raw_case(branch, c);
            unsigned long long pi = 0, po = 0;
            for (int i = 0; i < 8; ++i) {
                pi |= (static_cast<unsigned long long>(r.in[i]) & 0xFFull) << (8 * i);
                po |= (static_cast<unsigned long long>(r.out[i]) & 0xFFull) << (8 * i);
            }
            t.v[branch][c] = SqRawPacked{pi, po, r.nin, r.nout};
        }
    return t;
}
__device__ constexpr SqRawTable kSqRaw = sq_make_raw();
__host__ __device__ constexpr int sq_raw_elem(unsigned long long packed, int i) { return static_cast<int>(static_cast<signed char>((packed >> (8 * i)) & 0xFFull)); }

// test case `c` of `task`, already modulo'd and padded (host + device: eaz_subleq_test_vectors hands the same function to the CPU test that
// holds it against the oracle's generator; the device reads the packed table, the host evaluates sq_raw_case)
__host__ __device__ __forceinline__ void subleq_test_case(int task, int c, int W, uint16_t tin[8], uint16_t tout[8]) {
#ifdef __HIP_DEVICE_COMPILE__
    const SqRawPacked r = kSqRaw.v[sq_task_branch(task)][c];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        tin[i] = static_cast<uint16_t>(i < r.nin ? pymod(sq_raw_elem(r.in, i), W) : W);
        tout[i] = static_cast<uint16_t>(i < r.nout ? pymod(sq_raw_elem(r.out, i), W) : W);
    }
#else
    const SqRawCase r = sq_raw_case(sq_task_branch(task), c);
    for (int i = 0; i < 8; ++i) {
        tin[i] = static_cast<uint16_t>(i < r.nin ? pymod(r.in[i], W) : W);
        tout[i] = static_cast<uint16_t>(i < r.nout ? pymod(r.out[i], W) : W);
    }
#endif
}

// The test vectors of word_size 16, packed for the one-wavefront interpreter (env_kernels.hpp): [branch][case][0 = input, 1 = expected
// output], element i in byte i, W (= 16) beyond the valid ones -- sq_raw_case evaluated at compile time (its branch tree is ~300
// scalar instructions, ~2 k clocks of a lone wavefront per case, most of what an interpreter run used to cost).
struct Sq16CaseTable {
    unsigned long long v[6][3][2];
};
constexpr int sq16_pymod(int x) { return ((x % 16) + 16) % 16; }
constexpr Sq16CaseTable sq16_make_cases() {
    Sq16CaseTable t{};
    for (int branch = 0; branch < 6; ++branch)
        for (int c = 0; c < 3; ++c) {
            const SqRawCase r = sq_raw_case(branch, c);
            unsigned long long pi = 0, po = 0;
            for (int i = 0; i < 8; ++i) {
                pi |= static_cast<unsigned long long>(i < r.nin ? sq16_pymod(r.in[i]) : 16) << (8 * i);
                po |= static_cast<unsigned long long>(i < r.nout ? sq16_pymod(r.out[i]) : 16) << (8 * i);
            }
            t.v[branch][c][0] = pi;
            t.v[branch][c][1] = po;
        }
    return t;
}
__device__ constexpr Sq16CaseTable kSq16Cases = sq16_make_cases();
// subleq.py:156-395 simulate.  `mem` is a private, mutable copy of the program (W bytes, LDS or scratch).
// Returns correct (:394); in/out hold input_after / output_after on return.
// Eight uint16 values (test input / expected output / produced output) packed into two 64-bit registers: element i lives in
// bits 16*(i&3) of (i < 4 ? lo : hi).  Indexing by the run-time output cursor, the whole-array comparison of subleq.py:341 and the
// one-element shift of :333-338 become a handful of integer ops instead of dynamically indexed register arrays.
struct Pack8 {
    unsigned long long lo, hi;
};
__device__ __forceinline__ Pack8 pack8(const uint16_t v[8]) {
    Pack8 p{0ull, 0ull};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p.lo |= static_cast<unsigned long long>(v[i]) << (16 * i);
        p.hi |= static_cast<unsigned long long>(v[4 + i]) << (16 * i);
    }
    return p;
}
__device__ __forceinline__ void unpack8(const Pack8& p, uint16_t v[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = static_cast<uint16_t>(p.lo >> (16 * i));
        v[4 + i] = static_cast<uint16_t>(p.hi >> (16 * i));
    }
}
__device__ __forceinline__ bool pack8_eq(const Pack8& a, const Pack8& b) { return a.lo == b.lo && a.hi == b.hi; }
// subleq_test_case as two Pack8, from the compile-time table (no local arrays: the indexed uint16 arrays of subleq_test_case + pack8 cost the
// wavefront-per-environment kernel 112 bytes of scratch memory per lane)
__device__ __forceinline__ void subleq_test_case_packed(const int task, const int c, const int W, Pack8& pin, Pack8& pout) {
    const SqRawPacked r = kSqRaw.v[sq_task_branch(task)][c];
    pin = Pack8{0ull, 0ull}; pout = Pack8{0ull, 0ull};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned long long vi = static_cast<unsigned long long>(i < r.nin ? pymod(sq_raw_elem(r.in, i), W) : W) << (16 * (i & 3));
        const unsigned long long vo = static_cast<unsigned long long>(i < r.nout ? pymod(sq_raw_elem(r.out, i), W) : W) << (16 * (i & 3));
        if (i < 4) { pin.lo |= vi; pout.lo |= vo; } else { pin.hi |= vi; pout.hi |= vo; }
    }
}

// Program memory of one interpreter run: W bytes in LDS + W bytes for the periodicity snapshot (any word size; word_size 16, the
// size of the reference's experiments (slurm_experiment_submitter.py:196), has its own kernel: subleq16_wave_kernel).
struct MemLds {
    uint8_t* m;
    uint8_t* snap;
    int W;
    int ndiff = 0;  // bytes of m that differ from snap (meaningful once snapshot() has run): m == snap <=> ndiff == 0
    __device__ __forceinline__ int get(int i) const { return m[i]; }
    __device__ __forceinline__ int snap_get(int i) const { return snap[i]; }
    static constexpr bool kWordsInRange = false;  // bytes of a hand-made state may exceed W
    // old / s: the byte at i before the write and the snapshot's byte at i (the caller reads them with the operands, one LDS round trip).
    // The periodicity test used to compare the W bytes word by word whenever the cursor came back to the snapshot's: a counting loop
    // does that every iteration and differs only far up in memory -- 50 dependent LDS reads per Subleq cycle at word_size 256 (launches
    // of 600-700 us, profiles/r03_d_*); the running count makes the test one compare, exactly.
    __device__ __forceinline__ void set_if(bool cond, int i, int v, int old, int s) {
        if (cond) m[i] = static_cast<uint8_t>(v);
        ndiff += cond ? (static_cast<int>(v != s) - static_cast<int>(old != s)) : 0;
    }
    __device__ __forceinline__ void snapshot() {
        for (int i = 0; i < W; i += 4) *reinterpret_cast<uint32_t*>(snap + i) = *reinterpret_cast<const uint32_t*>(m + i);
        ndiff = 0;
    }
    __device__ __forceinline__ bool same() const { return ndiff == 0; }
};

// subleq.py:297-398 `simulate`.
//
// Exact early exit for non-terminating programs: the machine state (memory, cursor, remaining input, produced output, output
// cursor) is snapshotted at cycles 1, 2, 4, 8, ... (Brent); when the state equals the snapshot, the run is periodic with period
// p = cycles - snapshot cycle and nothing but the cycle counter changes from here on, so whole periods are skipped
// (cycles += floor((200 - cycles) / p) * p) and the last (200 - cycles) mod p cycles are interpreted normally.  The result is the
// state the reference reaches after exactly MAX_CYCLE_COUNT = 200 cycles (:19).
template <class Mem>
__device__ __forceinline__ bool subleq_simulate(int W, Mem& mem, uint16_t in_arr[8], const uint16_t tout_arr[8], uint16_t out_arr[8],
                                                int& bytes_used) {
    const int AMAX = W - 4, AIN = W - 3, AOUT = W - 2;  // :162-165
    const unsigned long long w4 = 0x0001000100010001ull * static_cast<unsigned long long>(W);
    Pack8 in = pack8(in_arr);
    const Pack8 tout = pack8(tout_arr);
    Pack8 out{w4, w4};  // :382
    int out_cursor = 0, cur = 0, cycles = 0;
    bool halt = false, error = false;
    // Brent snapshot.  `io_epoch` counts the cycles that consumed an input or produced an output: the input only ever shifts
    // forward and the output cursor only grows, so (in, out, out_cursor) equal their snapshot exactly when the epoch does.
    int io_epoch = 0, s_epoch = 0;
    int s_cur = -1, s_cycle = 0, next_snap = 1;
    bool skipped = false;
    bool all_eq = pack8_eq(out, tout);  // :341 `all(output == expected)`: changes only in cycles that write the output
    // The cycle body is written with selects instead of branches (lanes run different programs, and every divergent branch costs
    // the wavefront an exec-mask save/restore whether or not a lane takes it), with ONE wave-uniform branch: cycles in which no
    // lane of the wavefront touches the IO addresses -- the common case inside the long-running loops that bound the kernel --
    // skip the input shift, the output insert and both 128-bit comparisons.
    while (!error && !halt && cycles < 200) {  // :297-299, MAX_CYCLE_COUNT :19
        cycles += 1;
        const bool oob = cur + 2 >= W;  // :364-370: error, nothing else changes
        const bool ok = !oob;
        bytes_used = (ok && cur + 3 > bytes_used) ? cur + 3 : bytes_used;  // :311 (the skipped periods revisit cursors already counted)
        const int cc = oob ? 0 : cur;
        const int a = mem.get(cc), b = mem.get(cc + 1), c = mem.get(cc + 2);
        const bool a_mem = a <= AMAX, b_mem = b <= AMAX;
        const int ma = mem.get(a_mem ? a : 0), mb = mem.get(b_mem ? b : 0), sa = mem.snap_get(a_mem ? a : 0);
        const bool uses_in = (a == AIN) || (b == AIN), wr_out = a == AOUT;
        bool err = false;
        int value;
        if (__builtin_amdgcn_ballot_w64(ok && (uses_in || wr_out)) != 0) {
            // read_memory :187-232 (both reads see the pre-advance input state)
            const int in0 = static_cast<int>(in.lo & 0xFFFFu);
            const bool in_ok = in0 < W;
            const int ra = a_mem ? ma : ((a == AIN && in_ok) ? in0 : 0);
            const int rb = b_mem ? mb : ((b == AIN && in_ok) ? in0 : 0);
            const bool acc = ok && uses_in && in_ok;
            value = ra - rb;  // :322, see below
            value += value < 0 ? W : 0;
            if constexpr (!Mem::kWordsInRange) {
                if (value < 0 || value >= W) value = pymod(ra - rb, W);
            }
            // write_memory :243-295 (output port)
            const bool out_full = out_cursor >= 8;
            err = (uses_in && !in_ok) || (wr_out && out_full);
            const bool modified = ok && wr_out && !out_full;
            const int sh = 16 * (out_cursor & 3);
            const unsigned long long clr = ~(0xFFFFull << sh), ins = static_cast<unsigned long long>(value) << sh;
            const unsigned long long olo = (out.lo & clr) | ins, ohi = (out.hi & clr) | ins;
            out.lo = (modified && out_cursor < 4) ? olo : out.lo;
            out.hi = (modified && out_cursor >= 4) ? ohi : out.hi;
            out_cursor += modified ? 1 : 0;
            // :333-338 roll the input by one, pad with W
            const unsigned long long ilo = (in.lo >> 16) | (in.hi << 48), ihi = (in.hi >> 16) | (static_cast<unsigned long long>(W) << 48);
            in.lo = acc ? ilo : in.lo;
            in.hi = acc ? ihi : in.hi;
            io_epoch += (acc || modified) ? 1 : 0;
            all_eq = pack8_eq(out, tout);
            const int oc = (out_cursor - 1) & 7;  // -1 wraps to the last element (jnp negative index); masked by `modified`
            const unsigned long long diff = oc < 4 ? (out.lo ^ tout.lo) : (out.hi ^ tout.hi);
            err = err || (modified && ((diff >> (16 * (oc & 3))) & 0xFFFFull) != 0);  // :346-350
        } else {
            // no lane reads IN or writes OUT in this cycle: reads of the non-memory addresses return 0 (:187-232)
            const int ra = a_mem ? ma : 0, rb = b_mem ? mb : 0;
            value = ra - rb;
            value += value < 0 ? W : 0;
            if constexpr (!Mem::kWordsInRange) {
                if (value < 0 || value >= W) value = pymod(ra - rb, W);
            }
        }
        // :322 (ra - rb) mod W: words are < W, so one conditional add is the whole modulo; the general form (an integer division)
        // only runs for out-of-range bytes of a hand-made state.  write_memory :243-295 (memory)
        mem.set_if(ok && a_mem, a, value, ma, sa);
        const bool jump = (value == 0) || (2 * value >= W);  // :329
        cur = oob ? cur : (jump ? c : cur + 3);              // :330
        // :340-345 `(execute_jump & c > ADDRESS_MAX)` == `(execute_jump & c) > ADDRESS_MAX` (Python precedence): {0,1} > W-4
        halt = ok && ((((jump ? 1 : 0) & c) > AMAX) || all_eq);
        error = oob || err;
        // ---- periodicity (exact, see above) ----
        const bool watching = !skipped && !error && !halt;
        bool hit = (static_cast<int>(watching) & static_cast<int>(cur == s_cur) & static_cast<int>(io_epoch == s_epoch)) != 0;
        hit = (static_cast<int>(hit) & static_cast<int>(mem.same())) != 0;
        if (hit) {
            const int p = cycles - s_cycle;
            cycles += ((200 - cycles) / p) * p;
            skipped = true;
        }
        if (watching && !hit && cycles == next_snap) {
            mem.snapshot();
            s_cur = cur; s_epoch = io_epoch; s_cycle = cycles;
            next_snap *= 2;
        }
    }
    unpack8(in, in_arr);
    unpack8(out, out_arr);
    return !error && pack8_eq(out, tout);
}

// subleq.py:26-56 / :63-98 encoders: byte (row-word v, column j) of the observation
__device__ __forceinline__ uint8_t subleq_enc(const EnvDev& e, int v, int j) {
    if (e.binary) {
        if (j == e.obs_cols - 1) return static_cast<uint8_t>(v == e.W);               // :96
        const unsigned m = static_cast<unsigned>(pymod(v, e.W)) & 0xFFu;              // :90
        return j < 8 ? static_cast<uint8_t>((m >> j) & 1u) : static_cast<uint8_t>(0);  // :91
    }
    const int idx = v == e.W ? e.W : pymod(v, e.W);  // :52
    return static_cast<uint8_t>(j == idx);
}

// subleq.py:679-707 _observe: rows = memory | example_input | input_after | example_output | output_after
__device__ __forceinline__ int subleq_row_word(const EnvDev& e, const eaz_subleq_state* s, int row) {
    const int W = e.W;
    if (row < W) return s->memory[row];
    const int r = row - W;
    const int blk = r >> 3, i = r & 7;
    return blk == 0 ? s->ex_in[i] : (blk == 1 ? s->ex_in_after[i] : (blk == 2 ? s->ex_out[i] : s->ex_out_after[i]));
}

// generic observation byte i of the flattened observation
__device__ __forceinline__ uint8_t obs_byte(const EnvDev& e, const char* sp, int hot, int i) {
    if (e.kind == EAZ_ENV_DEEPSEA) return static_cast<uint8_t>(i == hot);
    if (e.kind == EAZ_ENV_CONNECT4) {  // [6,7,2]: plane 0 = stones of the colour to move, plane 1 = the other colour
        const eaz_connect4_state* c = reinterpret_cast<const eaz_connect4_state*>(sp);
        const int v = c->board[i >> 1];
        return static_cast<uint8_t>((i & 1) ? v == 1 - static_cast<int>(c->color) : v == static_cast<int>(c->color));
    }
    const eaz_subleq_state* s = reinterpret_cast<const eaz_subleq_state*>(sp);
    const int row = i / e.obs_cols, col = i - row * e.obs_cols;
    return subleq_enc(e, subleq_row_word(e, s, row), col);
}

// every state blob starts with {int32 step_count; int32 aux; float reward0; uint8 terminated, truncated, ..}
struct StateHeader {
    int32_t step_count, aux;
    float reward0;
    uint8_t terminated, truncated, x, y;
};
__device__ __forceinline__ bool state_is_done(const EnvDev& e, const char* sp) {
    const StateHeader* h = reinterpret_cast<const StateHeader*>(sp);
    return (h->terminated | h->truncated) != 0;
}

// ---------------------------------------------------------------------------------------------------------------
// connect_four: pgx.connect_four [EXT] (un-vendored, Pipfile:8) -- parity unpinned, game rules restated
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int c4_col_count(const eaz_connect4_state& s, int c) {
    int n = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) n += s.board[r * 7 + c] >= 0;
    return n;
}
__device__ __forceinline__ bool c4_won(const int8_t* b, int color) {
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 7; ++c) {
            if (b[r * 7 + c] != color) continue;
            if (c <= 3 && b[r * 7 + c + 1] == color && b[r * 7 + c + 2] == color && b[r * 7 + c + 3] == color) return true;
            if (r <= 2 && b[(r + 1) * 7 + c] == color && b[(r + 2) * 7 + c] == color && b[(r + 3) * 7 + c] == color) return true;
            if (r <= 2 && c <= 3 && b[(r + 1) * 7 + c + 1] == color && b[(r + 2) * 7 + c + 2] == color && b[(r + 3) * 7 + c + 3] == color) return true;
            if (r <= 2 && c >= 3 && b[(r + 1) * 7 + c - 1] == color && b[(r + 2) * 7 + c - 2] == color && b[(r + 3) * 7 + c - 3] == color) return true;
        }
    return false;
}
__device__ __forceinline__ void c4_init(eaz_connect4_state& s, Key key) {
    s.step_count = 0;
    s.current_player = static_cast<int32_t>(rbits(key, 0) & 1u);  // jnp.int32(jax.random.bernoulli(key))
    s.reward0 = 0.0f; s.reward1 = 0.0f;
    s.terminated = 0; s.truncated = 0; s.color = 0; s.pad = 0;
    s.winner = -1;
    for (int i = 0; i < 42; ++i) s.board[i] = -1;
    for (int i = 0; i < 14; ++i) s.pad2[i] = 0;
}
__device__ __forceinline__ bool c4_legal(const eaz_connect4_state& s, int a) { return s.terminated ? true : c4_col_count(s, a) < 6; }
// pgx.Env.step wrapper (A.1) around ConnectFour._step
__device__ __forceinline__ void c4_step(eaz_connect4_state& s, int action) {
    if (s.terminated | s.truncated) { s.reward0 = 0.0f; s.reward1 = 0.0f; return; }
    const int mover = s.current_player, color = s.color;
    const int n = c4_col_count(s, action);
    const bool illegal = n >= 6;
    s.step_count += 1;
    if (!illegal) s.board[(5 - n) * 7 + action] = static_cast<int8_t>(color);
    const bool won = c4_won(s.board, color);
    s.winner = won ? color : -1;
    s.color = static_cast<uint8_t>(1 - color);
    s.current_player = 1 - mover;
    bool full = true;
    for (int c = 0; c < 7; ++c) full = full && c4_col_count(s, c) == 6;
    s.terminated = (won || full) ? 1 : 0;
    s.reward0 = 0.0f; s.reward1 = 0.0f;
    if (won) { s.reward0 = mover == 0 ? 1.0f : -1.0f; s.reward1 = mover == 1 ? 1.0f : -1.0f; }
    if (illegal) {  // _step_with_illegal_action: mover -1, the other player +1, terminated
        s.reward0 = mover == 0 ? -1.0f : 1.0f;
        s.reward1 = mover == 1 ? -1.0f : 1.0f;
        s.terminated = 1;
    }
}

// ---- the same game on bitboards (device kernels): no indexed local array, hence no scratch memory ------------------------------
// Stones of colour c as bits of m[c], bit r * 7 + col (row 0 at the top, the blob's board order).  Results identical to c4_step /
// c4_init / c4_legal above (tests/test_gpu_golden.py::test_connect4_env_matches_oracle, the golden c4r case and the full-size C4 tests).
struct C4Bits {
    int32_t step_count, current_player, winner;
    float reward0, reward1;
    uint32_t terminated, truncated, color;
    unsigned long long m[2];
};
constexpr unsigned long long kC4Col0 = 1ull | (1ull << 7) | (1ull << 14) | (1ull << 21) | (1ull << 28) | (1ull << 35);  // column 0, rows 0..5
constexpr unsigned long long kC4All = (1ull << 42) - 1ull;
constexpr unsigned long long kC4ColLe3 = kC4Col0 | (kC4Col0 << 1) | (kC4Col0 << 2) | (kC4Col0 << 3);  // cells whose column is <= 3
constexpr unsigned long long kC4ColGe3 = (kC4Col0 << 3) | (kC4Col0 << 4) | (kC4Col0 << 5) | (kC4Col0 << 6);
// the 42 board bytes of a blob (words 6..16: board[0] is byte 24) -> bitboards
__device__ __forceinline__ void c4_bits_load(C4Bits& s, const uint4 (&v)[5]) {
    const uint32_t w[20] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w,
                            v[3].x, v[3].y, v[3].z, v[3].w, v[4].x, v[4].y, v[4].z, v[4].w};
    s.step_count = static_cast<int32_t>(w[0]); s.current_player = static_cast<int32_t>(w[1]);
    s.reward0 = __builtin_bit_cast(float, w[2]);
    s.terminated = w[3] & 0xFFu; s.truncated = (w[3] >> 8) & 0xFFu; s.color = (w[3] >> 16) & 0xFFu;
    s.reward1 = __builtin_bit_cast(float, w[4]);
    s.winner = static_cast<int32_t>(w[5]);
    unsigned long long m0 = 0ull, m1 = 0ull;
#pragma unroll
    for (int i = 0; i < 42; ++i) {
        const uint32_t byte = (w[6 + (i >> 2)] >> (8 * (i & 3))) & 0xFFu;
        m0 |= static_cast<unsigned long long>(byte == 0u) << i;
        m1 |= static_cast<unsigned long long>(byte == 1u) << i;
    }
    s.m[0] = m0; s.m[1] = m1;
}
// word j (0..19) of the 80-byte blob
__device__ __forceinline__ uint32_t c4_bits_word(const C4Bits& s, const int j) {
    if (j == 0) return static_cast<uint32_t>(s.step_count);
    if (j == 1) return static_cast<uint32_t>(s.current_player);
    if (j == 2) return __builtin_bit_cast(uint32_t, s.reward0);
    if (j == 3) return s.terminated | (s.truncated << 8) | (s.color << 16);
    if (j == 4) return __builtin_bit_cast(uint32_t, s.reward1);
    if (j == 5) return static_cast<uint32_t>(s.winner);
    if (j > 16) return 0u;
    uint32_t word = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * (j - 6) + k;
        uint32_t byte = 0u;  // pad2
        if (i < 42) byte = ((s.m[0] >> i) & 1ull) ? 0u : (((s.m[1] >> i) & 1ull) ? 1u : 0xFFu);
        word |= byte << (8 * k);
    }
    return word;
}
__device__ __forceinline__ bool c4_bits_won(const unsigned long long m) {
    const unsigned long long h = m & (m >> 1) & (m >> 2) & (m >> 3) & kC4ColLe3;     // right
    const unsigned long long v = m & (m >> 7) & (m >> 14) & (m >> 21);                // down
    const unsigned long long d = m & (m >> 8) & (m >> 16) & (m >> 24) & kC4ColLe3;    // down-right
    const unsigned long long a = m & (m >> 6) & (m >> 12) & (m >> 18) & kC4ColGe3;    // down-left
    return (h | v | d | a) != 0ull;
}
__device__ __forceinline__ void c4_bits_init(C4Bits& s, Key key) {
    s.step_count = 0;
    s.current_player = static_cast<int32_t>(rbits(key, 0) & 1u);  // jnp.int32(jax.random.bernoulli(key))
    s.reward0 = 0.0f; s.reward1 = 0.0f;
    s.terminated = 0u; s.truncated = 0u; s.color = 0u;
    s.winner = -1;
    s.m[0] = 0ull; s.m[1] = 0ull;
}
__device__ __forceinline__ int c4_bits_col_count(const C4Bits& s, const int a) { return __builtin_popcountll((s.m[0] | s.m[1]) & (kC4Col0 << a)); }
__device__ __forceinline__ bool c4_bits_legal(const C4Bits& s, const int a) { return s.terminated ? true : c4_bits_col_count(s, a) < 6; }
// pgx.Env.step wrapper (A.1) around ConnectFour._step: c4_step on bitboards
__device__ __forceinline__ void c4_bits_step(C4Bits& s, const int action) {
    if (s.terminated | s.truncated) { s.reward0 = 0.0f; s.reward1 = 0.0f; return; }
    const int mover = s.current_player;
    const uint32_t color = s.color;
    const int n = c4_bits_col_count(s, action);
    const bool illegal = n >= 6;
    s.step_count += 1;
    const unsigned long long stone = illegal ? 0ull : 1ull << ((5 - n) * 7 + action);
    const unsigned long long mine = (color ? s.m[1] : s.m[0]) | stone;
    if (color) s.m[1] = mine; else s.m[0] = mine;
    const bool won = c4_bits_won(mine);
    s.winner = won ? static_cast<int32_t>(color) : -1;
    s.color = 1u - color;
    s.current_player = 1 - mover;
    const bool full = (s.m[0] | s.m[1]) == kC4All;
    s.terminated = (won || full) ? 1u : 0u;
    s.reward0 = 0.0f; s.reward1 = 0.0f;
    if (won) { s.reward0 = mover == 0 ? 1.0f : -1.0f; s.reward1 = mover == 1 ? 1.0f : -1.0f; }
    if (illegal) {  // _step_with_illegal_action: mover -1, the other player +1, terminated
        s.reward0 = mover == 0 ? -1.0f : 1.0f;
        s.reward1 = mover == 1 ? -1.0f : 1.0f;
        s.terminated = 1u;
    }
}
// observation byte i of [6, 7, 2] (obs_byte): plane 0 = stones of the colour to move, plane 1 = the other colour
__device__ __forceinline__ uint32_t c4_bits_obs(const C4Bits& s, const int i) {
    const unsigned long long plane = (((i & 1) != 0) != (s.color != 0u)) ? s.m[1] : s.m[0];
    return static_cast<uint32_t>((plane >> (i >> 1)) & 1ull);
}

}  // namespace eaz
