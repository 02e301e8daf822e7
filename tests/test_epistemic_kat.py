"""Known-answer tree for the epistemic (beta > 0) search, worked by hand in exact arithmetic, so that the formulas in force are
written down independently of every implementation in this repository (emctx itself cannot be imported here: **parity unpinned**).

Setting: A = 2, S = 3 simulations, beta = 1, gamma = 0.5 (no terminal states), rescale_values = True, mixed value on,
value_scale 0.1, maxvisit_init 50, Gumbel g = [0.1, 0.0], root: logits [0, 0], value 0.5, variance 0.04.
Recurrent fn (reward, value, variance; logits [0, 0] everywhere):
    root -a0-> n1: r = 1, v = 0.2, var = 0.01        root -a1-> n2: r = 0, v = 0.8, var = 4.0        every deeper edge: r = 0, v = 0, var = 0.16
Considered-visit sequence for 2 actions, 3 simulations: 0, 0, 1.

DEFAULT variant (beta * sigma added BEFORE the min-max rescale; variance backup = running mean):
 sim 0  root: nothing visited, completed Q = v_mix = 0.5 for both, sigma = sqrt(0.04) = 0.2 -> [0.7, 0.7] -> rescaled [0, 0];
        score = g + logits + q = [0.1, 0.0] -> a0.  Expand n1.  Backup: G = 1 + 0.5 * 0.2 = 1.1, U = 0.25 * 0.01 = 0.0025;
        root value (0.5 + 1.1) / 2 = 0.8, root variance (0.04 + 0.0025) / 2 = 0.02125.
 sim 1  considered visit 0 -> only a1 (0 visits) qualifies -> a1.  Expand n2.  Backup: G = 0.5 * 0.8 = 0.4, U = 0.25 * 4 = 1.0;
        root value (0.8 * 2 + 0.4) / 3 = 2/3, root variance (0.02125 * 2 + 1.0) / 3 = 0.3475.
 sim 2  considered visit 1 -> both qualify.  Q = [1.1, 0.4], sigma = [sqrt(0.0025), sqrt(0.25 * 4)] = [0.05, 1.0]:
        Q + sigma = [1.15, 1.4] -> rescaled [0, 1] -> * (50 + 1) * 0.1 = [0, 5.1]; score [0.1, 5.1] -> a1.  At n2 (nothing visited):
        both completed Q equal -> a0 (ties to the lowest index).  Expand n3 below n2.  Backup: n2: G = 0, U = 0.25 * 0.16 = 0.04,
        n2 value (0.8 + 0) / 2 = 0.4, n2 variance (4 + 0.04) / 2 = 2.02; root: G = 0 + 0.5 * 0 = 0, U = 0.25 * 0.04 = 0.01,
        root value (2/3 * 3 + 0) / 4 = 0.5, root variance (0.3475 * 3 + 0.01) / 4 = 0.263125.
 final  root visits [1, 2]; action = the most visited = a1; Q(root) = [1.1, 0.5 * 0.4] = [1.1, 0.2],
        Q variance = [0.0025, 0.25 * 2.02] = [0.0025, 0.505].

bonus AFTER the rescale: sim 2 rescales Q = [1.1, 0.4] to [1, 0] first, adds sigma -> [1.05, 1.0] -> * 5.1 = [5.355, 5.1];
        score [5.455, 5.1] -> a0 instead: root visits [2, 1], root value (2 + 1) / 4 = 0.75 (G = 1 + 0.5 * 0 = 1).
variance backup = variance of the running mean: root variance 0.04 -> (0.04 * 1 + 0.0025) / 4 = 0.010625 -> (0.010625 * 4 + 1) / 9
        = 0.1158333.. -> (that * 9 + 0.01) / 16 = 0.06578125; n2 variance (4 * 1 + 0.04) / 4 = 1.01; same actions as the default.
"""
import numpy as np
import pytest

import mctx_numpy as M

GAMMA = 0.5
ROOT = dict(logits=[0.0, 0.0], value=0.5, var=0.04)
GUMBEL = [0.1, 0.0]


def _rf(state, action):
    """state = tuple of the actions from the root -> (reward, discount, logits, value, variance, next_state)"""
    path = state + (action,)
    if path == (0,):
        r, v, u = 1.0, 0.2, 0.01
    elif path == (1,):
        r, v, u = 0.0, 0.8, 4.0
    else:
        r, v, u = 0.0, 0.0, 0.16
    return r, GAMMA, np.zeros(2), v, u, path


EXPECT = {
    "default": dict(kw={}, root_visits=[1, 2], action=1, root_value=0.5, root_var=0.263125, n2_value=0.4, n2_var=2.02,
                    index=[[1, 2], [-1, -1], [3, -1], [-1, -1]], qvar_root=[0.0025, 0.505]),
    "bonus_after_rescale": dict(kw=dict(bonus_after_rescale=True), root_visits=[2, 1], action=0, root_value=0.75, root_var=0.263125,
                                n2_value=0.8, n2_var=4.0, index=[[1, 2], [3, -1], [-1, -1], [-1, -1]], qvar_root=[0.25 * 0.025, 1.0]),
    "var_of_mean": dict(kw=dict(var_backup=1), root_visits=[1, 2], action=1, root_value=0.5, root_var=0.06578125, n2_value=0.4,
                        n2_var=1.01, index=[[1, 2], [-1, -1], [3, -1], [-1, -1]], qvar_root=[0.0025, 0.25 * 1.01]),
}


@pytest.mark.parametrize("name", list(EXPECT))
def test_numpy_restatement_reproduces_the_hand_computed_tree(name):
    e = EXPECT[name]
    t, action, weights = M.search(3, 2, ROOT["logits"], ROOT["value"], ROOT["var"], (), _rf, GUMBEL, [False, False], 1.0, GAMMA, **e["kw"])
    assert t.children_visits[0].tolist() == e["root_visits"] and action == e["action"]
    assert t.children_index.tolist() == e["index"]
    assert t.node_values[0] == pytest.approx(e["root_value"], abs=1e-12) and t.node_var[0] == pytest.approx(e["root_var"], abs=1e-12)
    assert t.node_values[2] == pytest.approx(e["n2_value"], abs=1e-12) and t.node_var[2] == pytest.approx(e["n2_var"], abs=1e-12)
    np.testing.assert_allclose(t.qvar(0), e["qvar_root"], atol=1e-12)
    assert weights.sum() == pytest.approx(1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(EXPECT))
def test_engine_generic_recurrent_path_reproduces_the_hand_computed_tree(name):
    """The same tree through the device library: eaz_search_select / eaz_search_expand with the recurrent fn evaluated by the caller
    (the generic-recurrent-fn path of include/eaz.h), every batch element running the same case."""
    import torch

    import e_alphazero_amd as eaz
    from e_alphazero_amd import engine

    e = EXPECT[name]
    B, S, G = 5, 3, 4
    env = engine.EnvHandle.deepsea(G, np.zeros((G, G), np.uint8), device=0)   # only the tree is used: states are dummies
    net = engine.FCNet(env.ctx, G * G, 256, 2, eaz.params.init_fc_params(G * G, 256, 2, seed=0), eaz.params.init_hash_state())
    kw = dict(epistemic_bonus_after_rescale=e["kw"].get("bonus_after_rescale", False), epistemic_variance_backup=e["kw"].get("var_backup", 0))
    search = engine.Search(env, net, B, S, discount=GAMMA, **kw)
    f = lambda x: torch.tensor(np.broadcast_to(np.asarray(x, np.float32), (B,) + np.shape(x)).copy()).cuda()
    search.begin(f(ROOT["logits"]), f(ROOT["value"]), f(ROOT["var"]), env.init(B, key=0), f(1.0), None, f(GUMBEL))
    paths = {0: ()}
    for sim in range(S):
        parent, action, nxt = [x.cpu().numpy() for x in search.select(sim)]
        assert (parent == parent[0]).all() and (action == action[0]).all() and (nxt == nxt[0]).all()
        r, d, lg, v, u, ns = _rf(paths[int(parent[0])], int(action[0]))
        paths[int(nxt[0])] = ns
        search.expand(sim, f(r), f(0.0), f(d), f(lg), f(v), f(u))
    act, aw = search.finish()
    d = search.dump()
    assert d["children_visits"][0, 0].tolist() == e["root_visits"] and int(act[0]) == e["action"]
    assert d["children_index"][0].tolist()[:4] == e["index"]
    np.testing.assert_allclose([d["node_values"][0, 0], d["node_values_epistemic_variance"][0, 0]], [e["root_value"], e["root_var"]], atol=1e-6)
    np.testing.assert_allclose([d["node_values"][0, 2], d["node_values_epistemic_variance"][0, 2]], [e["n2_value"], e["n2_var"]], atol=1e-6)
    s = search.summary()
    np.testing.assert_allclose(s["qvalues_epistemic_variance"][0].cpu().numpy(), e["qvar_root"], atol=1e-6)
