"""Pins the oracle's environments on the known-answer material the reference files hold."""
import numpy as np
import pytest


# ---- DeepSea (src/envs/deep_sea.py) -----------------------------------------------------------------------------
@pytest.mark.parametrize("G", [4, 8, 32])
def test_deepsea_reward_iff_all_right(oracle, G):
    """deep_sea.py:62-78: reward 1 iff the submarine moved right G-1 times under the action map; terminates at step G-1."""
    rng = np.random.default_rng(G)
    am = (rng.random((G, G)) < 0.5).astype(np.uint8)
    env = oracle.OracleEnv.deepsea(G, am)
    B = 64
    st = env.init(B)
    obs0 = env.observe(st).reshape(B, G, G)
    assert obs0[:, 0, 0].all() and obs0.sum() == B  # deep_sea.py:55-57
    # batch 0: always the "right" action under the map; others random
    pos = np.zeros(B, np.int64)
    for t in range(G - 1):
        a = rng.integers(0, 2, B).astype(np.int32)
        right_action = np.where(am[t, pos] != 0, 0, 1)  # (action==0) xor flip -> left; so right = action != 0 xor flip
        a[0] = right_action[0]
        st = env.step(st, a)
        f = env.fields(st)
        moved_right = (a == 0) ^ (am[t, pos] != 0) ^ True
        pos = np.clip(pos + np.where(moved_right, 1, -1), 0, G - 1)
        assert (f["step_count"] == t + 1).all()
        assert (f["terminated"] == (t + 1 >= G - 1)).all()
        obs = env.observe(st).reshape(B, G, G)
        assert obs.sum() == B and obs[np.arange(B), min(t + 1, G - 1), pos].all()
    f = env.fields(st)
    assert f["terminated"].all()
    assert np.array_equal(f["rewards"], (pos == G - 1).astype(np.float32))
    assert f["rewards"][0] == 1.0
    # pgx wrapper: stepping a terminal state is a no-op with zero rewards (A.1)
    st2 = env.step(st, np.zeros(B, np.int32))
    f2 = env.fields(st2)
    assert (f2["rewards"] == 0).all() and f2["terminated"].all() and (f2["step_count"] == G - 1).all()
    assert np.array_equal(env.observe(st2), env.observe(st))
    # auto_reset (selfplay.py:66-71): terminal states are replaced by init states
    st3 = env.step(st, np.zeros(B, np.int32), autoreset=True)
    assert np.array_equal(st3, env.init(B))


def test_deepsea_no_action_map_is_plain_left_right(oracle):
    env = oracle.OracleEnv.deepsea(4, None)
    st = env.init(1)
    for a, p in ((1, 1), (1, 2), (0, 1)):
        st = env.step(st, np.array([a], np.int32))
        assert np.flatnonzero(env.observe(st)[0])[0] % 4 == p


def test_deepsea_all_states_sweep(oracle):
    """main.py:114-124 (debug_deep_sea): every one-hot grid is a distinct network input with a distinct hash index."""
    G = 8
    allst = np.zeros((G * G, G * G), np.uint8)
    allst[np.arange(G * G), np.arange(G * G)] = 1
    idx = np.empty(G * G, np.uint32)
    oracle.lib().eazo_xxhash_indices(allst.ctypes.data, G * G, G * G, 24, idx.ctypes.data)
    assert len(set(idx.tolist())) == G * G


# ---- Subleq encoders (src/envs/subleq.py docstrings) -------------------------------------------------------------
def test_subleq_one_hot_docstring_example(oracle):
    """subleq.py:30-41: word_size = 8, [1, 3, 5, -1, 8]."""
    words = np.array([1, 3, 5, -1, 8], np.int32)
    rows = np.zeros((5, 9), np.uint8)
    assert oracle.lib().eazo_subleq_encode_raw(8, 0, words.ctypes.data, 5, rows.ctypes.data) == 9
    expected = np.array([[0, 1, 0, 0, 0, 0, 0, 0, 0],
                         [0, 0, 0, 1, 0, 0, 0, 0, 0],
                         [0, 0, 0, 0, 0, 1, 0, 0, 0],
                         [0, 0, 0, 0, 0, 0, 0, 1, 0],
                         [0, 0, 0, 0, 0, 0, 0, 0, 1]], np.uint8)
    assert np.array_equal(rows, expected)


def test_subleq_binary_docstring_example(oracle):
    """subleq.py:67-78: word_size = 16, [1, 3, 5, -1, 16]."""
    words = np.array([1, 3, 5, -1, 16], np.int32)
    rows = np.zeros((5, 5), np.uint8)
    assert oracle.lib().eazo_subleq_encode_raw(16, 1, words.ctypes.data, 5, rows.ctypes.data) == 5
    expected = np.array([[1, 0, 0, 0, 0],
                         [1, 1, 0, 0, 0],
                         [1, 0, 1, 0, 0],
                         [1, 1, 1, 1, 0],
                         [0, 0, 0, 0, 1]], np.uint8)
    assert np.array_equal(rows, expected)


def test_subleq_binary_width(oracle):
    for W, width in ((16, 5), (32, 6), (256, 9), (20, 6)):  # subleq.py:59-60
        env = oracle.OracleEnv.subleq(W, (1,), True)
        assert env.obs_shape == (W + 32, width)
        assert oracle.OracleEnv.subleq(W, (1,), False).obs_shape == (W + 32, W + 1)


# ---- Subleq test tables (src/envs/subleq.py:406-454, pad rule :101-107) ------------------------------------------
def _cases(oracle, task, W):
    tin = np.zeros((3, 8), np.int32)
    tout = np.zeros((3, 8), np.int32)
    oracle.lib().eazo_subleq_test_cases(task, W, tin.ctypes.data, tout.ctypes.data)
    return tin, tout


def _pad(v, W):
    out = np.full(8, W, np.int64)
    out[: len(v)] = np.mod(v, W)
    return out


@pytest.mark.parametrize("W", [16, 256])
def test_subleq_test_tables(oracle, W):
    np1 = [list(range(1, 8)), [5, 4, 4, 5, 1, 2, 3, 1], [1, 1, 6, 2, 4, 4, 5, 3]]
    neg = [list(range(-4, 4)), list(range(1, 8)), [0, -1, 2, -3, 4]]
    arith_in = [[1, 1, 5, 4, 0, -3], [2, 3, 0, 0, -1, -2], [1, 2, 3, 4, 4, 3, 2, 1]]
    tables = {
        1: (np1, [[-x for x in r] for r in np1]),
        2: (neg, [[-x for x in r] for r in neg]),
        3: (neg, neg),
        4: (arith_in, [[0, 1, 3], [-1, 0, 1], [-1, -1, 1, 1]]),
        5: (arith_in, [[2, 9, -3], [5, 0, -3], [3, 7, 7, 3]]),
        6: ([[1, 2, 2, 3, -7, 4], [-1, -5, 5, 2, 0, 1], [0, 0, 1, 1, -3, 3]], [[2, 6, -28], [5, 10, 0], [0, 1, -9]]),
    }
    for task, (ins, outs) in tables.items():
        tin, tout = _cases(oracle, task, W)
        for c in range(3):
            assert np.array_equal(tin[c], _pad(ins[c], W)), (task, c)
            assert np.array_equal(tout[c], _pad(outs[c], W)), (task, c)
    # lax.switch clamps: every task id >= 6 (MAXIMUM ... SUMMATION) gets the multiplication table (:459-501)
    t6 = _cases(oracle, 6, W)
    for task in (7, 9, 12, 14):
        t = _cases(oracle, task, W)
        assert np.array_equal(t[0], t6[0]) and np.array_equal(t[1], t6[1])


# ---- Subleq interpreter / env known answers ---------------------------------------------------------------------
def _simulate(oracle, W, mem, tin, tout):
    m = np.zeros(W, np.int32)
    m[: len(mem)] = mem
    ia, oa, info = np.zeros(8, np.int32), np.zeros(8, np.int32), np.zeros(3, np.int32)
    oracle.lib().eazo_subleq_simulate(W, m.ctypes.data, np.ascontiguousarray(tin, np.int32).ctypes.data,
                                      np.ascontiguousarray(tout, np.int32).ctypes.data, ia.ctypes.data, oa.ctypes.data, info.ctypes.data)
    return ia, oa, dict(bytes_used=int(info[0]), cycles=int(info[1]), correct=bool(info[2]))


def test_subleq_empty_program_loops_to_cycle_limit(oracle):
    """All-zero memory is `subleq 0 0 0`: mem[0] -= mem[0] -> 0 -> jump to 0, forever: 200 cycles (subleq.py:19,297-299)."""
    tin, tout = _cases(oracle, 1, 16)
    ia, oa, info = _simulate(oracle, 16, [], tin[0], tout[0])
    assert info == dict(bytes_used=3, cycles=200, correct=False)
    assert np.array_equal(ia, tin[0]) and (oa == 16).all()


def test_subleq_negation_positive_two_word_solution(oracle):
    """SubleqTask comment (subleq.py:111): NEGATION_POSITIVE is solvable with 2 written words.
    `subleq OUT IN 0` = [W-2, W-3, 0]: out <- 0 - in; the result is >= W/2 so it jumps back to 0; halts on full match."""
    for W in (16, 256):
        tin, tout = _cases(oracle, 1, W)
        for c in range(3):
            ia, oa, info = _simulate(oracle, W, [W - 2, W - 3], tin[c], tout[c])
            assert info["correct"] and np.array_equal(oa, tout[c])
            n = int((tin[c] < W).sum())
            assert info["cycles"] == n and (ia == W).all()  # every input consumed, one instruction each
        env = oracle.OracleEnv.subleq(W, (1,), True)
        st = env.init(1)
        assert env.fields(st)["rewards"][0] == 0.0  # the empty program does not solve the task (subleq.py:631-633)
        st = env.step(st, np.array([W - 2], np.int32))
        f = env.fields(st)
        assert f["rewards"][0] == 0.0 and not f["terminated"][0]
        st = env.step(st, np.array([W - 3], np.int32))
        f = env.fields(st)
        assert f["rewards"][0] == 1.0 and not f["terminated"][0]  # solved: reward now, termination on the NEXT step
        st = env.step(st, np.array([0], np.int32))
        f = env.fields(st)
        assert f["terminated"][0] and f["rewards"][0] == 0.0  # subleq.py:671-673


def test_subleq_min_solution_length_is_two(oracle):
    """Exhaustive over all 1- and 2-word programs at W = 16: no 1-word solution; the 2-word solutions include [14, 13]."""
    W = 16
    env = oracle.OracleEnv.subleq(W, (1,), True)
    st0 = env.init(W)
    st1 = env.step(st0, np.arange(W, dtype=np.int32))
    assert (env.fields(st1)["rewards"] == 0).all()
    solved = []
    for a0 in range(W):
        s = env.step(env.init(W), np.full(W, a0, np.int32))
        s = env.step(s, np.arange(W, dtype=np.int32))
        r = env.fields(s)["rewards"]
        solved += [(a0, a1) for a1 in range(W) if r[a1] == 1.0]
    assert (14, 13) in solved
    assert all(a0 == 14 for a0, _ in solved)  # only writes to OUT can produce output


def test_subleq_episode_length_and_wrong_output_error(oracle):
    W = 16
    env = oracle.OracleEnv.subleq(W, (1,), True)
    st = env.init(2)
    for t in range(W - 4):  # steps with _step_count 1..12 write memory[0..11]
        st = env.step(st, np.array([3, 5], np.int32))
        assert not env.fields(st)["terminated"].any()
    st = env.step(st, np.array([3, 5], np.int32))  # _step_count 13 >= W-3 -> terminated, reward 0 (subleq.py:671-673)
    f = env.fields(st)
    assert f["terminated"].all() and (f["rewards"] == 0).all() and (f["step_count"] == W - 3).all()
    # first wrong output is an error (subleq.py:346-350): [14, 0, 0] writes 0 - mem[0] = 2 != expected 15
    tin, tout = _cases(oracle, 1, W)
    ia, oa, info = _simulate(oracle, W, [14], tin[0], tout[0])
    assert not info["correct"] and info["cycles"] == 1 and oa[0] == 2


def test_subleq_halt_jump_is_dead_code(oracle):
    """subleq.py:340-345: `(execute_jump & c > ADDRESS_MAX)` parses as `(execute_jump & c) > ADDRESS_MAX`; jumping to the
    HALT address therefore does not halt -- the next fetch is out of bounds and raises an error instead (:364-370)."""
    W = 16
    tin, tout = _cases(oracle, 1, W)
    ia, oa, info = _simulate(oracle, W, [0, 0, 15], tin[0], tout[0])  # mem[0]-=mem[0] -> 0 -> jump to 15
    assert info["cycles"] == 2 and not info["correct"]


def test_subleq_observation_layout(oracle):
    """subleq.py:697-705: memory | example_input | input_after | example_output | output_after."""
    W = 16
    env = oracle.OracleEnv.subleq(W, (1,), True)
    st = env.step(env.step(env.init(1), np.array([14], np.int32)), np.array([13], np.int32))
    obs = env.observe(st).reshape(W + 32, 5)
    tin, tout = _cases(oracle, 1, W)

    def enc(words):
        rows = np.zeros((len(words), 5), np.uint8)
        w = np.ascontiguousarray(words, np.int32)
        oracle.lib().eazo_subleq_encode_raw(W, 1, w.ctypes.data, len(w), rows.ctypes.data)
        return rows

    mem = np.zeros(W, np.int32)
    mem[:2] = (14, 13)
    assert np.array_equal(obs[:W], enc(mem))
    assert np.array_equal(obs[W:W + 8], enc(tin[0]))
    assert np.array_equal(obs[W + 8:W + 16], enc(np.full(8, W)))  # all input consumed
    assert np.array_equal(obs[W + 16:W + 24], enc(tout[0]))
    assert np.array_equal(obs[W + 24:W + 32], enc(tout[0]))        # output after == expected output


def test_subleq_lowest_bytes_known_answers(oracle):
    """Subleq(reward_fn=lowest_bytes), subleq.py:540-542: solved / (1 + bytes_used), bytes_used = highest cursor + 3 of any executed cycle
    (:311), maximum over the three test runs (:530).  Hand-worked at W = 16 (IN = 13, OUT = 14):
      [14, 13]            `subleq OUT IN 0` runs at cursor 0 only                       -> bytes_used = 3, reward 1/4
      [0, 0, 3, 14, 13]   `subleq 0 0 3` (mem[0] = 0, jump 3) then `subleq OUT IN 0` at cursor 3, back through cursor 0
                                                                                        -> bytes_used = 6, reward 1/7
    and every shorter prefix of the second program is unsolved (reward 0 under either function)."""
    W = 16
    for prog, want in (([14, 13], np.float32(1.0) / np.float32(4.0)), ([0, 0, 3, 14, 13], np.float32(1.0) / np.float32(7.0))):
        for fn, solved_reward in (("lowest_bytes", want), ("solved_or_not", np.float32(1.0))):
            env = oracle.OracleEnv.subleq(W, (1,), True, reward_fn=fn)
            st = env.init(1)
            for t, word in enumerate(prog):
                st = env.step(st, np.array([word], np.int32))
                r = env.fields(st)["rewards"][0]
                assert r == (solved_reward if t == len(prog) - 1 else 0.0), (prog, fn, t, r)
    with pytest.raises((ValueError, KeyError)):
        oracle.OracleEnv.subleq(W, (1,), True, reward_fn="fewest_cycles")
