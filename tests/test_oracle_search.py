"""Pins for the oracle's Gumbel-MuZero search: sequential-halving known answers, structural invariants of
mctx trees, and an independent float64 NumPy restatement (tests/mctx_numpy.py).  **parity unpinned** against real emctx."""
import numpy as np
import pytest

import mctx_numpy as M
from helpers import deepsea_setup, oracle_deepsea, random_states_deepsea, subleq_setup


def _rle(seq):
    out = []
    for v in seq:
        if out and out[-1][0] == v:
            out[-1][1] += 1
        else:
            out.append([int(v), 1])
    return [tuple(x) for x in out]


def test_sequential_halving_known_answers(oracle):
    """SURVEY.md A.4 run-length known answers of get_sequence_of_considered_visits."""
    assert _rle(oracle.considered_visits(2, 16)) == [(i, 2) for i in range(8)]
    assert _rle(oracle.considered_visits(2, 64)) == [(i, 2) for i in range(32)]
    assert _rle(oracle.considered_visits(16, 64)) == [(0, 16), (1, 8), (2, 8), (3, 4), (4, 4), (5, 4), (6, 4)] + [(i, 2) for i in range(7, 15)]
    exp7 = [(i, 7) for i in range(9)] + [(i, 3) for i in range(9, 31)] + [(i, 2) for i in range(31, 66)] + [(66, 1)]
    assert _rle(oracle.considered_visits(7, 200)) == exp7
    assert list(oracle.considered_visits(1, 5)) == [0, 1, 2, 3, 4] and list(oracle.considered_visits(0, 3)) == [0, 1, 2]
    for m in range(0, 17):
        for S in (1, 5, 16, 32, 64, 200):
            assert list(oracle.considered_visits(m, S)) == M.considered_visits_sequence(m, S)


def _run(oracle, eaz, G, B, S, beta_max, seed, **variant):
    am, params, state = deepsea_setup(eaz, G, seed=seed, fill=0.5)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    blobs = random_states_deepsea(G, B, seed=seed + 1)
    ex, xp, v, u, _ = onet.forward(oenv.observe(blobs))
    beta = (beta_max * np.linspace(0, 1, B)).astype(np.float32)
    g = oracle.gumbel((seed, 1), B * 2).reshape(B, 2)
    act, aw, summ, tree = oracle.search(oenv, onet, blobs, S, xp if beta_max > 0 else ex, v, u, beta, g, discount=0.997,
                                        use_exploration_logits=beta_max > 0, **variant)
    return dict(env=oenv, net=onet, blobs=blobs, logits=xp if beta_max > 0 else ex, v=v, u=u, beta=beta, g=g, act=act, aw=aw,
                summ=summ, tree=tree.dump())


def test_tree_invariants_deepsea(oracle, eaz):
    G, B, S = 8, 48, 16
    r = _run(oracle, eaz, G, B, S, 0.0, 7)
    d = r["tree"]
    assert (d["node_visits"][:, 0] == S + 1).all()
    assert (d["children_visits"][:, 0].sum(-1) == S).all()
    # two valid root actions, 16 simulations: sequential halving gives each root action exactly 8 visits (A.4)
    assert (d["children_visits"][:, 0] == 8).all()
    for b in range(B):
        for n in range(1, S + 1):  # node n was first expanded at simulation n-1 (A.3)
            p, a = d["parents"][b, n], d["action_from_parent"][b, n]
            assert p >= 0 and d["children_index"][b, p, a] == n
            assert d["children_visits"][b, p, a] == d["node_visits"][b, n]
            assert d["children_values"][b, p, a] == d["node_values"][b, n]
        assert (d["node_visits"][b, 1:] >= 1).all()
        ci = d["children_index"][b]
        assert sorted(ci[ci >= 0].tolist()) == list(range(1, S + 1))
    np.testing.assert_allclose(r["aw"].sum(-1), 1.0, atol=1e-6)
    assert np.array_equal(r["summ"]["visit_counts"], d["children_visits"][:, 0].astype(np.float32))


def test_deepsea32_64_sims_give_32_visits_each(oracle, eaz):
    r = _run(oracle, eaz, 32, 8, 64, 0.0, 11)
    assert (r["tree"]["children_visits"][:, 0] == 32).all()  # m=2, S=64


def _recurrent(oenv, onet, discount, explore):
    def fn(state, action):
        ns = oenv.step(state[None], np.array([action], np.int32))
        ex, xp, v, u, _ = onet.forward(oenv.observe(ns))
        f = oenv.fields(ns)
        term = bool(f["terminated"][0])
        lg = (xp if explore else ex)[0].astype(np.float64)
        lg = lg - lg.max()
        return (float(f["rewards"][0]), 0.0 if term else discount, lg, 0.0 if term else float(v[0]), 0.0 if term else float(u[0]), ns[0])
    return fn


@pytest.mark.parametrize("beta_max,variant", [
    (0.0, {}), (1.0, {}),
    # the two unpinned epistemic choices (include/eaz.h eaz_search_opts): the C oracle and the NumPy restatement follow the same switches
    (1.0, dict(bonus_after_rescale=True)), (1.0, dict(var_backup=1)), (2.0, dict(bonus_after_rescale=True, var_backup=1)),
    (0.0, dict(var_backup=1)),
])
def test_against_numpy_restatement_deepsea(oracle, eaz, beta_max, variant):
    G, B, S = 8, 12, 16
    r = _run(oracle, eaz, G, B, S, beta_max, 21, **variant)
    fn = _recurrent(r["env"], r["net"], float(np.float32(0.997)), beta_max > 0)
    for b in range(B):
        t, a, w = M.search(S, 2, r["logits"][b], float(r["v"][b]), float(r["u"][b]), r["blobs"][b], fn, r["g"][b],
                           [False, False], float(r["beta"][b]), float(np.float32(0.997)), **variant)
        d = r["tree"]
        assert np.array_equal(t.children_index, d["children_index"][b]), b
        assert np.array_equal(t.children_visits, d["children_visits"][b]), b
        assert np.array_equal(t.parents, d["parents"][b]) and np.array_equal(t.action_from_parent, d["action_from_parent"][b])
        np.testing.assert_allclose(d["node_values"][b], t.node_values, atol=1e-5)
        np.testing.assert_allclose(d["node_values_epistemic_variance"][b], t.node_var, atol=1e-5)
        np.testing.assert_allclose(d["children_values"][b], t.children_values, atol=1e-5)
        assert a == r["act"][b]
        np.testing.assert_allclose(r["aw"][b], w, atol=1e-5)


def test_against_numpy_restatement_subleq(oracle, eaz):
    """A = 16 actions: exercises m = 16 sequential halving, the butterfly reductions and large fan-out."""
    W, B, S = 16, 6, 32
    dims, params, st = subleq_setup(eaz, W, seed=2, fill=0.3)
    oenv = oracle.OracleEnv.subleq(W, (1,), True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=st[eaz.params.HASH_MODULE]["binary_set"])
    states = oenv.step(oenv.init(B), np.array([14, 13, 0, 5, 14, 3], np.int32))
    states = oenv.step(states, np.array([1, 13, 2, 6, 13, 3], np.int32))  # two envs are now solved: terminal children appear
    ex, xp, v, u, _ = onet.forward(oenv.observe(states))
    beta = np.linspace(0, 1, B).astype(np.float32)
    g = oracle.gumbel((3, 3), B * W).reshape(B, W)
    invalid = np.zeros((B, W), np.uint8)
    invalid[0, 3:9] = 1  # masked root actions (connect-four style) even though Subleq itself has none
    act, aw, summ, tree = oracle.search(oenv, onet, states, S, xp, v, u, beta, g, invalid=invalid, discount=0.97,
                                        use_exploration_logits=True)
    d = tree.dump()
    fn = _recurrent(oenv, onet, float(np.float32(0.97)), True)
    for b in range(B):
        t, a, w = M.search(S, W, xp[b], float(v[b]), float(u[b]), states[b], fn, g[b], invalid[b], float(beta[b]),
                           float(np.float32(0.97)))
        assert np.array_equal(t.children_index, d["children_index"][b]), b
        assert np.array_equal(t.children_visits, d["children_visits"][b]), b
        np.testing.assert_allclose(d["node_values"][b], t.node_values, atol=1e-5)
        np.testing.assert_allclose(d["node_values_epistemic_variance"][b], t.node_var, atol=1e-5)
        assert a == act[b]
        np.testing.assert_allclose(aw[b], w, atol=1e-5)
    assert (d["children_visits"][0, 0, 3:9] == 0).all() and not (3 <= act[0] < 9)
