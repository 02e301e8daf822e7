"""Pins for the C4 pieces of the oracle: connect_four (pgx [EXT], parity unpinned: game-rule invariants only) and the
residual epistemic net (src/network/resnet.py) against a float64 NumPy restatement."""
import numpy as np
import pytest

C4 = np.dtype([("step_count", "<i4"), ("current_player", "<i4"), ("reward0", "<f4"), ("terminated", "u1"), ("truncated", "u1"),
               ("color", "u1"), ("pad", "u1"), ("reward1", "<f4"), ("winner", "<i4"), ("board", "i1", 42), ("pad2", "u1", 14)])
assert C4.itemsize == 80


def _view(states):
    return np.ascontiguousarray(states).view(C4).reshape(states.shape[:-1])


def _play(oracle, env, moves):
    st = env.init(1, key=(1, 0))
    out = [st]
    for m in moves:
        st = env.step(st, np.array([m], np.int32))
        out.append(st)
    return out


def test_connect4_basics(oracle):
    env = oracle.OracleEnv.connect4()
    assert env.num_actions == 7 and env.obs_dim == 84 and env.state_bytes == 80
    st = env.init(64, key=(3, 0))
    v = _view(st)
    assert (v["board"] == -1).all() and (v["winner"] == -1).all() and (v["color"] == 0).all()
    assert set(v["current_player"].tolist()) == {0, 1}  # random player identities (bernoulli at init)
    assert env.observe(st).sum() == 0
    # gravity: the first stone of column 3 lands on the bottom row (row 5), the second one above it
    s = _play(oracle, env, [3, 3])
    b = _view(s[2])[0]["board"].reshape(6, 7)
    assert b[5, 3] == 0 and b[4, 3] == 1 and (b >= 0).sum() == 2
    # observation is from the perspective of the player to move: plane 0 = own stones
    obs = env.observe(s[1]).reshape(6, 7, 2)  # color 1 to move, one color-0 stone on the board
    assert obs[5, 3, 1] == 1 and obs[..., 0].sum() == 0
    obs = env.observe(s[2]).reshape(6, 7, 2)  # color 0 to move again
    assert obs[5, 3, 0] == 1 and obs[4, 3, 1] == 1


@pytest.mark.parametrize("moves,winner_moves_first", [
    ([0, 1, 0, 1, 0, 1, 0], True),             # vertical four in column 0
    ([0, 0, 1, 1, 2, 2, 3], True),             # horizontal four on the bottom row
    ([0, 1, 1, 2, 2, 3, 2, 3, 3, 6, 3], True),  # diagonal / for the first player
    ([6, 0, 0, 1, 1, 2, 1, 2, 2, 3, 2, 3, 3, 5, 3], False),  # the same diagonal shifted for the SECOND player? (checked by rules below)
])
def test_connect4_wins(oracle, moves, winner_moves_first):
    env = oracle.OracleEnv.connect4()
    s = _play(oracle, env, moves)
    first_term = next((i for i, x in enumerate(s) if _view(x)[0]["terminated"]), len(s) - 1)
    s = s[: first_term + 1]  # later moves are terminal no-ops
    v0, vl = _view(s[0])[0], _view(s[-1])[0]
    first_player = int(v0["current_player"])
    board = vl["board"].reshape(6, 7)

    def four(color):
        for r in range(6):
            for c in range(7):
                for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
                    cells = [(r + k * dr, c + k * dc) for k in range(4)]
                    if all(0 <= rr < 6 and 0 <= cc < 7 and board[rr, cc] == color for rr, cc in cells):
                        return True
        return False

    won0, won1 = four(0), four(1)
    assert bool(vl["terminated"]) == (won0 or won1)
    if won0 or won1:
        wc = 0 if won0 else 1
        assert vl["winner"] == wc
        winner_player = first_player if wc == 0 else 1 - first_player
        r = [vl["reward0"], vl["reward1"]]
        assert r[winner_player] == 1.0 and r[1 - winner_player] == -1.0
        for prev in s[:-1]:
            assert not _view(prev)[0]["terminated"]
        # terminal no-op: zero rewards, same board; legal mask all-true at terminal
        nxt = env.step(s[-1], np.array([0], np.int32))
        vn = _view(nxt)[0]
        assert vn["reward0"] == 0 and vn["reward1"] == 0 and np.array_equal(vn["board"], vl["board"]) and vn["terminated"]


def test_connect4_illegal_move_and_draw(oracle):
    env = oracle.OracleEnv.connect4()
    s = _play(oracle, env, [2] * 6)
    v = _view(s[-1])[0]
    assert not v["terminated"] and (v["board"].reshape(6, 7)[:, 2] >= 0).all()
    mover = int(v["current_player"])
    bad = env.step(s[-1], np.array([2], np.int32))  # column full -> illegal: mover -1, opponent +1, terminated (pgx A.1)
    vb = _view(bad)[0]
    r = [vb["reward0"], vb["reward1"]]
    assert vb["terminated"] and r[mover] == -1.0 and r[1 - mover] == 1.0
    # a full board without a four is a draw: fill columns in an order that avoids lines of four
    order = [0, 1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0,   2, 3, 2, 3, 2, 3, 3, 2, 3, 2, 3, 2,   4, 5, 4, 5, 4, 5, 5, 4, 5, 4, 5, 4,   6, 6, 6, 6, 6, 6]
    s = _play(oracle, env, order)
    v = _view(s[-1])[0]
    if v["winner"] < 0:
        assert v["terminated"] and v["reward0"] == 0 and v["reward1"] == 0 and (v["board"] >= 0).all()


def _np_resnet(eaz, params, state, obs, A, nb, seen):
    P = eaz.params.RESNET_PREFIX
    x = obs.reshape(-1, 6, 7, 2).astype(np.float64)

    def conv(x, name):
        w, b = params[name]["w"].astype(np.float64), params[name]["b"].astype(np.float64)
        kh = w.shape[0]
        pad = kh // 2
        xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
        y = np.zeros(x.shape[:3] + (w.shape[3],))
        for i in range(kh):
            for j in range(kh):
                y += xp[:, i:i + 6, j:j + 7, :] @ w[i, j]
        return y + b

    def bn(x, name):
        p = params[name]
        m, v = state[name + "/~/mean_ema"]["average"].astype(np.float64), state[name + "/~/var_ema"]["average"].astype(np.float64)
        return (x - m) * (p["scale"] / np.sqrt(v + 1e-5)) + p["offset"]

    relu = lambda a: np.maximum(a, 0)
    x1 = conv(x, f"{P}/conv2_d")
    for b in range(nb):
        i = x1
        t = conv(relu(bn(x1, f"{P}/block_{b}/batch_norm")), f"{P}/block_{b}/conv2_d")
        x1 = conv(relu(bn(t, f"{P}/block_{b}/batch_norm_1")), f"{P}/block_{b}/conv2_d_1") + i
    x1 = relu(bn(x1, f"{P}/batch_norm"))

    def lin(a, name):
        return a @ params[name]["w"].astype(np.float64) + params[name]["b"]

    def head(ci, bi):
        return relu(bn(conv(x1, f"{P}/conv2_d_{ci}"), f"{P}/batch_norm_{bi}")).reshape(x.shape[0], -1)

    pol = lin(head(1, 1), f"{P}/linear")
    xpl = lin(head(2, 2), f"{P}/linear_1")
    v = np.tanh(lin(relu(lin(head(3, 3), f"{P}/linear_2")), f"{P}/linear_3"))[:, 0]
    u = 0.5 * (np.tanh(lin(relu(lin(head(4, 4), f"{P}/linear_4")), f"{P}/linear_5"))[:, 0] + 1)
    nov = (~seen).astype(np.float64)
    return pol, xpl, v, np.maximum(nov, u), nov


def test_resnet_forward_against_float64(oracle, eaz):
    nb, A, B = 2, 7, 9
    params, state = eaz.params.init_resnet_params(6, 7, 2, A, 64, nb, seed=1)
    for k in params:  # non-zero biases
        if "b" in params[k]:
            params[k]["b"] = (0.1 * np.random.default_rng(2).standard_normal(params[k]["b"].shape)).astype(np.float32)
    bs = eaz.params.init_hash_state(fill=0.5, seed=3)[eaz.params.HASH_MODULE]["binary_set"]
    net = oracle.OracleNet.resnet(6, 7, 2, A, eaz.params.resnet_array_list(params, state, nb), 64, nb, bitset=bs)
    env = oracle.OracleEnv.connect4()
    st, _ = oracle.random_play(env, B, 9, key=(4, 0))
    obs = env.observe(st)
    assert 0 < obs.sum() < obs.size
    ex, xp, v, u, nov = net.forward(obs)
    idx = np.empty(B, np.uint32)
    oracle.lib().eazo_xxhash_indices(obs.ctypes.data, B, 84, 24, idx.ctypes.data)
    seen = np.empty(B, np.uint8)
    oracle.lib().eazo_hash_seen(bs.ctypes.data, idx.ctypes.data, B, seen.ctypes.data)
    r = _np_resnet(eaz, params, state, obs, A, nb, seen.astype(bool))
    np.testing.assert_allclose(ex, r[0], atol=2e-5)
    np.testing.assert_allclose(xp, r[1], atol=2e-5)
    np.testing.assert_allclose(v, r[2], atol=1e-5)
    np.testing.assert_allclose(u, r[3], atol=1e-5)
    assert np.array_equal(nov, r[4])


def test_connect4_selfplay_two_player_search(oracle, eaz):
    """Two-player discount (-gamma, context.py:143): a search from a position with a winning move must pick it."""
    nb, A = 1, 7
    params, state = eaz.params.init_resnet_params(6, 7, 2, A, 64, nb, seed=5)
    net = oracle.OracleNet.resnet(6, 7, 2, A, eaz.params.resnet_array_list(params, state, nb), 64, nb)
    env = oracle.OracleEnv.connect4()
    s = _play(oracle, env, [0, 6, 0, 6, 0, 5])  # color 0 has three in column 0 and is to move
    st = np.repeat(s[-1], 4, axis=0)
    ex, xp, v, u, _ = net.forward(env.observe(st))
    g = np.zeros((4, 7), np.float32)
    act, aw, summ, tree = oracle.search(env, net, st, 64, ex, v, u, np.zeros(4, np.float32), g, discount=0.997, two_player=True)
    assert (act == 0).all()
    d = tree.dump()
    assert (d["children_rewards"][:, 0, 0] == 1.0).all() and (d["children_discounts"][:, 0, 0] == 0.0).all()
    assert (summ["qvalues"][:, 0] == 1.0).all()
