"""GPU parity at BASELINE.json's FULL sizes (C2 DeepSea-32 B=4096 S=64; C3 Subleq-16 B=4096 S=64; C4 connect_four B=8192 S=200).

The oracle cannot run a whole full-size batch in seconds, so these tests use what the domain offers:
  * trees / environments are independent: the full-batch device result, restricted to a random SUBSET of environments, must equal
    the oracle run on that subset alone (same states, same explicit Gumbel noise);
  * size-independent invariants of the search tree (visit-count bookkeeping, topology) on every one of the B trees;
  * determinism (same key -> bit-identical output; captured hipGraph == plain launches).
Bar: integers bit-exact, floats within 1e-5 (north-star tolerance).
"""
import numpy as np
import pytest

from helpers import deepsea_setup, oracle_deepsea, subleq_device_to_oracle, subleq_setup

pytestmark = pytest.mark.gpu
ATOL = 1e-5
FLOATS = ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance")


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available(), "GPU tests need a ROCm device"
    return t


def _check_subset(out, ref, idx, state_cmp=None):
    assert np.array_equal(out["action"][:, idx].cpu().numpy(), ref["action"])
    dev_states = out["state"][:, idx].cpu().numpy()
    assert np.array_equal(dev_states if state_cmp is None else state_cmp(dev_states), ref["state"])
    for k in FLOATS:
        np.testing.assert_allclose(out[k][:, idx].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)


def test_c2_selfplay_subset_matches_oracle_and_is_deterministic(oracle, eaz, torch):
    from e_alphazero_amd import engine

    G, B, S, T, NSUB = 32, 4096, 64, 34, 24  # T > 31: every episode terminates and auto-resets inside the window
    am, params, state = deepsea_setup(eaz, G, seed=0, fill=0.5)
    env = engine.EnvHandle.deepsea(G, am, device=0)
    net = engine.FCNet(env.ctx, G * G, 256, 2, params, state)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    search = engine.Search(env, net, B, S, discount=0.997)
    gum = oracle.gumbel((5, 6), T * B * 2).reshape(T, B, 2)
    g_dev = torch.from_numpy(gum).cuda()
    st = env.init(B, key=0)
    out = search.selfplay(st, T, key=17, gumbel=g_dev)
    st2 = env.init(B, key=0)
    out2 = search.selfplay(st2, T, key=17, gumbel=g_dev)
    st3 = env.init(B, key=0)
    out3 = search.selfplay(st3, T, key=17, gumbel=g_dev, use_graph=False)
    for k in out:
        assert torch.equal(out[k], out2[k]) and torch.equal(out[k], out3[k]), k
    idx = np.sort(np.random.default_rng(1).choice(B, NSUB, replace=False))
    ost = oenv.init(NSUB)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(17, 0), gumbel=np.ascontiguousarray(gum[:, idx]))
    _check_subset(out, ref, idx)
    assert np.array_equal(st[idx].cpu().numpy(), ost)
    # the window contains whole episodes: rewards only at the last row, and some environment reached the treasure or none did --
    # either way every stored state is a legal DeepSea state
    sc = out["state"].cpu().numpy()[..., 0:4].copy().view(np.int32)[..., 0]
    assert sc.min() >= 0 and sc.max() <= G - 1


def test_c2_tree_invariants_on_every_tree(oracle, eaz, torch):
    from e_alphazero_amd import engine

    G, B, S = 32, 4096, 64
    am, params, state = deepsea_setup(eaz, G, seed=0, fill=0.5)
    env = engine.EnvHandle.deepsea(G, am, device=0)
    net = engine.FCNet(env.ctx, G * G, 256, 2, params, state)
    search = engine.Search(env, net, B, S, discount=0.997)
    st = env.init(B, key=0)
    for t in range(5):  # a few real steps in, so the roots are spread over the grid
        st = env.step(st, torch.randint(0, 2, (B,), dtype=torch.int32, device="cuda"))
    ex, xp, v, u, _ = net.forward(env.observe(st))
    g = torch.from_numpy(oracle.gumbel((8, 9), B * 2).reshape(B, 2)).cuda()
    search.begin(ex, v, u, st, torch.zeros(B, device="cuda"), None, g)
    search.run()
    act, aw = search.finish()
    d = search.dump()
    N = S + 1
    nv, ci, cv, par, afp = d["node_visits"], d["children_index"], d["children_visits"], d["parents"], d["action_from_parent"]
    assert (nv[:, 0] == S + 1).all() and (cv[:, 0].sum(1) == S).all()          # root: initial visit + one per simulation
    assert (nv[:, 1:] >= 1).all()                                              # max_depth = S: every simulation expands a new node
    assert (nv == 1 + cv.sum(2)).all()                                         # visits of a node = its own expansion + its children's
    b = np.arange(B)[:, None].repeat(N - 1, 1)
    n = np.arange(1, N)[None, :].repeat(B, 0)
    assert (par[:, 1:] < n).all() and (par[:, 1:] >= 0).all()                  # nodes are created in simulation order
    assert (ci[b, par[:, 1:], afp[:, 1:]] == n).all()                          # parent/child links agree
    assert ((ci == -1) == (cv == 0)).all()                                     # unvisited <=> no child
    assert (nv[b, par[:, 1:]] > nv[:, 1:]).all()                               # a parent is visited more often than its child
    s = search.summary()
    assert torch.equal(s["visit_counts"].sum(1), torch.full((B,), float(S), device="cuda"))
    np.testing.assert_allclose(aw.sum(1).cpu().numpy(), 1.0, atol=1e-5)
    assert ((act >= 0) & (act < 2)).all()
    # the chosen action is one of the most visited ones (Gumbel sequential halving ends on the max-visit set)
    vc = s["visit_counts"].cpu().numpy()
    assert (vc[np.arange(B), act.cpu().numpy()] == vc.max(1)).all()


@pytest.mark.parametrize("flags,constants_first,full_obs_hash", [({}, False, False), (dict(directed_exploration=True, exploration_beta=1.0), False, False),
                                                                 (dict(directed_exploration=True, exploration_beta=1.0), True, False), ({}, True, False),
                                                                 ({}, False, True)],
                         ids=["beta0", "beta1", "beta1-constants-first", "beta0-constants-first", "beta0-full-obs-hash"])
def test_c3_selfplay_subset_matches_oracle(oracle, eaz, torch, flags, constants_first, full_obs_hash):
    """T = 16 real steps: Subleq-16 episodes end at _step_count >= 13 (subleq.py:671-673), so the window holds searches from terminated roots
    and the auto-reset inside selfplay for every environment; beta1 = the reference's experiment setting (directed exploration, beta = 1,
    slurm_experiment_submitter.py:203,209) at full size."""
    from e_alphazero_amd import engine

    W, B, S, T, NSUB = 16, 4096, 64, 16, 16
    dims, params, state = subleq_setup(eaz, W, seed=0, fill=0.5, hash_io=not full_obs_hash)  # full_obs_hash: subleq_hash_only_io = False (config.py:30)
    env = engine.EnvHandle.subleq(W, (1,), True, device=0)
    order = eaz.params.subleq_constants_first(W, True) if constants_first else None  # the numerics switch of include/eaz.h eaz_net_set_input_order
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"], input_order=order)
    oenv = oracle.OracleEnv.subleq(W, (1,), True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    onet.set_input_order(order)
    search = engine.Search(env, net, B, S, discount=0.97, exploration=bool(flags))
    assert search.launch_info()["one_launch"], "C3 is expected on the one-launch kernel (both layer-1 input orders)"
    gum = oracle.gumbel((3, 4), T * B * W).reshape(T, B, W)
    st = env.init(B, key=0)
    out = search.selfplay(st, T, key=23, gumbel=torch.from_numpy(gum).cuda(), **flags)
    idx = np.sort(np.random.default_rng(2).choice(B, NSUB, replace=False))
    if flags:
        idx[-1] = B - 1  # beta * linspace(0, 1, B)[b] is per environment (selfplay.py:105): the oracle gets the subset's own betas below
    ost = oenv.init(NSUB)  # one task: init does not depend on the key
    okw = dict(flags)
    if flags:
        okw["beta_per_env"] = (flags["exploration_beta"] * np.linspace(0, 1, B)[idx]).astype(np.float32)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(23, 0), gumbel=np.ascontiguousarray(gum[:, idx]), discount=0.97, **okw)
    _check_subset(out, ref, idx, state_cmp=lambda s: subleq_device_to_oracle(s, W))
    assert np.array_equal(subleq_device_to_oracle(st[idx].cpu().numpy(), W), ost)
    sc = ref["state"].view(np.uint8)[..., 0:4].copy().view(np.int32)[..., 0]
    # EVERY environment of the subset crosses an episode boundary inside the window (a terminated-root search, then the auto-reset)
    assert (np.diff(sc.astype(np.int64), axis=0) < 0).any(axis=0).all(), "an environment never restarted inside the window"


@pytest.mark.parametrize("wave", [1, 0], ids=["wave-per-env", "lane-per-env"])
def test_subleq256_selfplay_subset_matches_oracle(oracle, eaz, torch, wave):
    """Both Subleq step kernels of the per-simulation path (EAZ_TUNE_SUBLEQ_WAVE: subleq_wave_kernel, the default, and subleq_step_kernel).
    The reference's code-default word size (main.py:159: word_size = 256 -> 256 actions, 2592 observation bytes, 532 KB of tree statistics per
    environment) at C3's batch and simulation count, on the per-simulation kernels (tree_step_kernel<64, 4>, subleq_step_kernel, the 16- /
    64-row forward) with sub-batches on their own streams: a random subset of the batch against the oracle over 3 real steps."""
    from e_alphazero_amd import engine

    W, B, S, T, NSUB = 256, 4096, 64, 3, 6
    engine.tune(subleq_wave=wave)
    request_restore = lambda: engine.tune(subleq_wave=1)  # noqa: E731
    dims, params, state = subleq_setup(eaz, W, seed=1, fill=0.5)
    env = engine.EnvHandle.subleq(W, (1,), True, device=0)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    oenv = oracle.OracleEnv.subleq(W, (1,), True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    search = engine.Search(env, net, B, S, discount=0.97)
    idx = np.sort(np.random.default_rng(4).choice(B, NSUB, replace=False))
    idx[0], idx[-1] = 0, B - 1
    # root noise only for the subset's rows is drawn from the oracle's generator; the other rows get an independent stream
    gum = np.random.default_rng(5).gumbel(size=(T, B, W)).astype(np.float32)
    gum[:, idx] = oracle.gumbel((6, 7), T * NSUB * W).reshape(T, NSUB, W)
    st = env.init(B, key=0)
    try:
        out = search.selfplay(st, T, key=31, gumbel=torch.from_numpy(gum).cuda())
        torch.cuda.synchronize()
    finally:
        request_restore()
    ost = oenv.init(NSUB)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(31, 0), gumbel=np.ascontiguousarray(gum[:, idx]), discount=0.97)
    _check_subset(out, ref, idx, state_cmp=lambda s: subleq_device_to_oracle(s, W))
    assert np.array_equal(subleq_device_to_oracle(st[idx].cpu().numpy(), W), ost)


@pytest.mark.parametrize("flags", [{}, dict(directed_exploration=True, exploration_beta=1.0)], ids=["beta0", "beta1"])
def test_c4_selfplay_subset_matches_oracle(oracle, eaz, torch, flags):
    """beta1: directed exploration at full size (exploration policy head in the search, beta * linspace(0, 1, B)[b] per environment, variance
    loads in the tree kernels) -- the subset's own betas go to the oracle."""
    from e_alphazero_amd import engine

    B, S, T, NSUB, nb = 8192, 200, 1, 6, 5
    params, state = eaz.params.init_resnet_params(6, 7, 2, 7, 64, nb, seed=0)
    bs = eaz.params.init_hash_state(fill=0.3, seed=0)[eaz.params.HASH_MODULE]["binary_set"]
    state[eaz.params.RESNET_HASH_MODULE] = {"binary_set": bs}
    env = engine.EnvHandle.connect4(device=0)
    net = engine.ResNet(env.ctx, 6, 7, 2, 7, params, state, num_blocks=nb)
    oenv = oracle.OracleEnv.connect4()
    onet = oracle.OracleNet.resnet(6, 7, 2, 7, eaz.params.resnet_array_list(params, state, nb), 64, nb, bitset=bs)
    search = engine.Search(env, net, B, S, discount=0.997, two_players_game=True, exploration=bool(flags))
    st = env.init(B, key=5)
    rng = np.random.default_rng(3)
    for t in range(6):  # mid-game positions
        st = env.step(st, torch.from_numpy(rng.integers(0, 7, B).astype(np.int32)).cuda())
    idx = np.sort(rng.choice(B, NSUB, replace=False))
    okw = dict(flags)
    if flags:
        idx[-1] = B - 1
        okw["beta_per_env"] = (flags["exploration_beta"] * np.linspace(0, 1, B)[idx]).astype(np.float32)
    ost = st[idx].cpu().numpy().copy()  # identical 80-byte layout on both sides
    gum = oracle.gumbel((7, 8), T * B * 7).reshape(T, B, 7)
    out = search.selfplay(st, T, key=29, gumbel=torch.from_numpy(gum).cuda(), **flags)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(29, 0), gumbel=np.ascontiguousarray(gum[:, idx]), two_player=True, **okw)
    _check_subset(out, ref, idx)
    s = search.summary()
    live = ~torch.from_numpy(oenv.fields(ref["state"][0])["terminated"]).bool()
    assert torch.equal(s["visit_counts"].sum(1)[idx][live], torch.full((int(live.sum()),), float(S), device="cuda"))
