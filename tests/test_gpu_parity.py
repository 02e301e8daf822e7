"""GPU parity: the HIP engine (through the C ABI of include/eaz.h) against the CPU oracle on identical seeded inputs.

Bar: integer / byte / index results bit-exact; float results within 1e-5 (north-star tolerance), written below.
"""
import numpy as np
import pytest

from helpers import deepsea_setup, oracle_deepsea, random_states_deepsea

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("kernel_variant")]
ATOL = 1e-5

INT_FIELDS = ["node_visits", "parents", "action_from_parent", "children_index", "children_visits"]
FLOAT_FIELDS = ["raw_values", "node_values", "children_prior_logits", "children_values", "children_rewards",
                "children_discounts", "raw_values_epistemic_variance", "node_values_epistemic_variance",
                "children_values_epistemic_variance", "children_rewards_epistemic_variance"]


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available(), "GPU tests need a ROCm device"
    return t


def _engine(eaz, G, am, params, state):
    from e_alphazero_amd import engine

    env = engine.EnvHandle.deepsea(G, am, device=0)
    net = engine.FCNet(env.ctx, G * G, 256, 2, params, state)
    return engine, env, net


@pytest.mark.parametrize("G", [4, 8, 32])
def test_deepsea_env_step_observe(oracle, eaz, torch, G):
    am, params, state = deepsea_setup(eaz, G, seed=G)
    engine, env, _ = _engine(eaz, G, am, params, state)
    oenv = oracle.OracleEnv.deepsea(G, am)
    B = 257
    st = env.init(B, key=3)
    ost = oenv.init(B)
    assert np.array_equal(st.cpu().numpy(), ost)
    rng = np.random.default_rng(G)
    for t in range(2 * G + 3):  # runs past termination: terminal no-op (step) and reset (autoreset) branches
        a = rng.integers(0, 2, size=B).astype(np.int32)
        auto = t % 2 == 1 or t > G
        st = env.step(st, torch.from_numpy(a).cuda(), key=t, autoreset=auto)
        ost = oenv.step(ost, a, key=(t, 0), autoreset=auto)
        assert np.array_equal(st.cpu().numpy(), ost), f"state mismatch at t={t}"
        assert np.array_equal(env.observe(st).cpu().numpy(), oenv.observe(ost))
        f, of = env.fields(st), oenv.fields(ost)
        assert np.array_equal(f["terminated"].cpu().numpy(), of["terminated"])
        assert np.array_equal(f["rewards"].cpu().numpy()[:, 0], of["rewards"])


@pytest.mark.parametrize("G,B", [(8, 64), (32, 300), (6, 33)])
def test_fc_forward_and_hash(oracle, eaz, torch, G, B):
    am, params, state = deepsea_setup(eaz, G, seed=1, fill=0.5)
    engine, env, net = _engine(eaz, G, am, params, state)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    if (G * G) % 4:
        pytest.skip("hash input must be a multiple of 4 (hashes.py:210)")
    blobs = random_states_deepsea(G, B, seed=2)
    obs = oenv.observe(blobs)
    # multi-hot rows as well: the kernel must be a real dense GEMM, not a one-hot shortcut
    obs[::3] = (np.random.default_rng(5).random(obs[::3].shape) < 0.3).astype(np.uint8)
    ref = onet.forward(obs)
    out = net.forward(torch.from_numpy(obs).cuda())
    names = ["exploitation_logits", "exploration_logits", "value", "ube", "novelty"]
    for n, a, b in zip(names, out, ref):
        a = a.cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=ATOL, err_msg=n)
    # the f32 MFMA chain is specified to be the k-ordered fmaf chain of the oracle: report bit equality
    exact = all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(out, ref))
    print(f"fc forward bit-exact vs oracle: {exact}")
    seen, idx = net.hash_lookup(torch.from_numpy(obs).cuda())
    oidx = np.empty(B, np.uint32)
    oracle.lib().eazo_xxhash_indices(obs.ctypes.data, B, G * G, 24, oidx.ctypes.data)
    assert np.array_equal(idx.cpu().numpy().view(np.uint32), oidx)


def _compare_tree(search, otree, tag):
    d, od = search.dump(), otree.dump()
    for k in INT_FIELDS:
        assert np.array_equal(d[k], od[k]), f"{tag}: integer field {k} differs"
    for k in FLOAT_FIELDS:
        np.testing.assert_allclose(d[k], od[k], rtol=0, atol=ATOL, err_msg=f"{tag}: {k}")
    return all(np.array_equal(d[k], od[k]) for k in FLOAT_FIELDS)


@pytest.mark.parametrize("G,B,S,beta_max,fill", [(8, 64, 16, 0.0, 0.0), (8, 70, 16, 1.0, 0.5), (32, 128, 64, 0.0, 0.5)])
def test_search_tree_per_simulation(oracle, eaz, torch, G, B, S, beta_max, fill):
    am, params, state = deepsea_setup(eaz, G, seed=3, fill=fill)
    engine, env, net = _engine(eaz, G, am, params, state)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    blobs = random_states_deepsea(G, B, seed=4)
    obs = oenv.observe(blobs)
    ex, xp, v, u, _ = onet.forward(obs)  # the ORACLE's root outputs are injected into both searches
    beta = (beta_max * np.linspace(0, 1, B)).astype(np.float32)
    g = oracle.gumbel((11, 12), B * 2).reshape(B, 2)
    kw = dict(discount=0.997, use_exploration_logits=beta_max > 0)
    otree = oracle.OracleTree(oenv, onet, B, S)
    otree.begin(ex, v, u, blobs, beta, None, g, **kw)
    search = engine.Search(env, net, B, S, discount=0.997, exploration=beta_max > 0)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    search.begin(cu(ex), cu(v), cu(u), cu(blobs), cu(beta), None, cu(g))
    exact = True
    for sim in range(S):
        otree.sim(sim)
        search.simulate(sim)
        if sim in (0, 1, S // 2, S - 1):
            # the device expands simulation `sim` lazily (fused with the next select); finish() flushes it
            pass
    act, aw = search.finish()
    oact, oaw = otree.finish()
    exact &= _compare_tree(search, otree, "final")
    assert np.array_equal(act.cpu().numpy(), oact)
    np.testing.assert_allclose(aw.cpu().numpy(), oaw, rtol=0, atol=ATOL)
    s, os_ = search.summary(), otree.summary()
    for k in os_:
        np.testing.assert_allclose(s[k].cpu().numpy(), os_[k], rtol=0, atol=ATOL, err_msg=k)
    assert np.array_equal(s["visit_counts"].cpu().numpy(), os_["visit_counts"])
    print(f"tree float fields bit-exact vs oracle: {exact}")


@pytest.mark.parametrize("max_depth,rescale,mixed", [(3, True, True), (1, True, True), (5, False, True), (4, True, False)])
def test_search_depth_limited_and_qtransform_options(oracle, eaz, torch, max_depth, rescale, mixed):
    """max_depth < S (mctx `max_depth`): simulations that reach the limit re-expand an existing node (visit count + 1, embedding
    rewritten) -- the path on which the fused DeepSea step writes into a slot that is already occupied -- together with the
    qtransform switches (rescale_values, use_mixed_value)."""
    G, B, S = 8, 96, 24
    am, params, state = deepsea_setup(eaz, G, seed=8, fill=0.3)
    engine, env, net = _engine(eaz, G, am, params, state)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    blobs = random_states_deepsea(G, B, seed=9)
    ex, xp, v, u, _ = onet.forward(oenv.observe(blobs))
    beta = np.zeros(B, np.float32)
    g = oracle.gumbel((21, 22), B * 2).reshape(B, 2)
    otree = oracle.OracleTree(oenv, onet, B, S)
    otree.begin(ex, v, u, blobs, beta, None, g, discount=0.997, rescale=rescale, use_mixed=mixed, max_depth=max_depth)
    search = engine.Search(env, net, B, S, discount=0.997, rescale_values=rescale, use_mixed_value=mixed, max_depth=max_depth)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    search.begin(cu(ex), cu(v), cu(u), cu(blobs), cu(beta), None, cu(g))
    for sim in range(S):
        otree.sim(sim)
        search.simulate(sim)
    act, aw = search.finish()
    oact, oaw = otree.finish()
    _compare_tree(search, otree, f"max_depth={max_depth}")
    assert np.array_equal(search.field("embeddings").cpu().numpy(), otree.field("embeddings"))
    assert np.array_equal(act.cpu().numpy(), oact)
    np.testing.assert_allclose(aw.cpu().numpy(), oaw, rtol=0, atol=ATOL)
    nv = search.dump()["node_visits"]
    assert (nv[:, 1:] > 1).any(), "the depth limit never re-expanded a node: the test does not cover what it claims"


@pytest.mark.parametrize("G,B,S,beta_max,opts,masked", [
    (8, 64, 16, 0.0, {}, False),
    (8, 70, 16, 1.0, {}, False),                                       # epistemic twin + ragged last workgroup (70 = 4 * 16 + 6)
    (32, 300, 64, 0.0, {}, False),                                     # C2's tree shape
    (32, 48, 64, 0.7, dict(max_depth=6), False),                       # depth limit: re-expansion of existing nodes
    (8, 33, 24, 0.0, dict(max_depth=1), False),
    (8, 40, 40, 0.0, dict(rescale=False), True),                       # invalid root actions (incl. rows with every action invalid)
    (8, 40, 24, 0.5, dict(use_mixed=False, max_considered=1), True),
    (6, 17, 8, 0.0, dict(two_player=True), False),                     # negated discount (context.py:143)
    (8, 20, 70, 0.0, {}, False),                                       # S = 70: the trees of a workgroup no longer fit into LDS
    # the two unpinned epistemic choices (include/eaz.h eaz_search_opts), oracle and device behind the same switches
    (8, 50, 24, 2.0, dict(bonus_after_rescale=True), False),
    (8, 50, 24, 1.0, dict(var_backup=1), False),
    (32, 40, 64, 1.5, dict(bonus_after_rescale=True, var_backup=1, max_depth=9), False),
    # the smallest searches: one, two and three simulations (trees of 2-4 nodes, one-entry considered-visit sequences)
    (8, 20, 1, 0.0, {}, False),
    (8, 20, 2, 0.5, {}, True),
    (8, 20, 3, 0.0, dict(max_considered=1), False),
])
def test_search_run_whole_tree(oracle, eaz, torch, kernel_variant, G, B, S, beta_max, opts, masked):
    """begin -> run (all S simulations in one call) -> finish: every tree field, the embeddings, action and weights against the
    oracle.  Under the `persist` variant eligible cases are ONE launch of search_persist_kernel (trees in LDS); the other variants
    (and S = 70) loop the per-simulation kernels -- all of them must agree with the oracle."""
    am, params, state = deepsea_setup(eaz, G, seed=13, fill=0.4)
    engine, env, net = _engine(eaz, G, am, params, state)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    blobs = random_states_deepsea(G, B, seed=14)
    ex, xp, v, u, _ = onet.forward(oenv.observe(blobs))
    beta = (beta_max * np.linspace(0, 1, B)).astype(np.float32)
    g = oracle.gumbel((31, 32), B * 2).reshape(B, 2)
    invalid = None
    if masked:
        invalid = (np.random.default_rng(3).random((B, 2)) < 0.3).astype(np.uint8)
        invalid[0] = 1
    explore = beta_max > 0
    logits = xp if explore else ex
    okw = dict(discount=0.997, use_exploration_logits=explore, rescale=opts.get("rescale", True), use_mixed=opts.get("use_mixed", True),
               max_considered=opts.get("max_considered", 16), max_depth=opts.get("max_depth", 0), two_player=opts.get("two_player", False),
               bonus_after_rescale=opts.get("bonus_after_rescale", False), var_backup=opts.get("var_backup", 0))
    otree = oracle.OracleTree(oenv, onet, B, S)
    otree.begin(logits, v, u, blobs, beta, invalid, g, **okw)
    for sim in range(S):
        otree.sim(sim)
    search = engine.Search(env, net, B, S, discount=0.997, exploration=explore, rescale_values=okw["rescale"], use_mixed_value=okw["use_mixed"],
                           max_num_considered_actions=okw["max_considered"], max_depth=okw["max_depth"] or None,
                           two_players_game=okw["two_player"], epistemic_bonus_after_rescale=okw["bonus_after_rescale"],
                           epistemic_variance_backup=okw["var_backup"])
    cu = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    info = search.launch_info()
    # DeepSea: EAZ_TUNE_PERSIST 1 and 2 both mean the one-launch kernel; S = 70: the trees of a workgroup no longer fit into LDS
    assert info["one_launch"] == (kernel_variant in ("persist", "persist-step") and S <= 64), (info, S)
    for rep in range(2):  # the second search reuses the (dirty) tree buffers of the first
        search.begin(cu(logits), cu(v), cu(u), cu(blobs), cu(beta), cu(invalid), cu(g))
        search.profile(True)
        search.profile_read()
        search.run()
        torch.cuda.synchronize()
        prof = search.profile_read()
        search.profile(False)
        assert (prof["search"]["launches"] == 1 and prof["tree"]["launches"] == 0) == info["one_launch"], (prof, info)
        act, aw = search.finish()
        oact, oaw = otree.finish()
        exact = _compare_tree(search, otree, f"run rep={rep}")
        emb, oemb = search.field("embeddings").cpu().numpy(), otree.field("embeddings")
        visited = search.dump()["node_visits"] > 0
        assert np.array_equal(emb[visited], oemb[visited]), "embeddings of visited nodes differ"
        assert np.array_equal(act.cpu().numpy(), oact)
        np.testing.assert_allclose(aw.cpu().numpy(), oaw, rtol=0, atol=ATOL)
        s_, os_ = search.summary(), otree.summary()
        for k in os_:
            np.testing.assert_allclose(s_[k].cpu().numpy(), os_[k], rtol=0, atol=ATOL, err_msg=k)
    print(f"run(): tree float fields bit-exact vs oracle: {exact}")


@pytest.mark.parametrize("B,S,beta_max,opts,masked", [
    (64, 16, 0.0, {}, False),
    (37, 32, 1.0, {}, False),                                          # epistemic twin; 37 = 2 * 16 + 5: mirror trees in the last workgroup
    (300, 64, 0.0, {}, False),                                         # C3's tree shape
    (5, 68, 0.5, {}, False),                                           # the largest tree the LDS record set holds (one ragged workgroup)
    (5, 69, 0.0, {}, False),                                           # first size beyond it: subleq16_search_kernel / per-simulation kernels
    (40, 24, 0.0, dict(max_depth=4), False),                           # depth limit: re-expansion of existing nodes
    (33, 24, 0.0, dict(max_depth=1), False),
    (40, 40, 0.0, dict(rescale=False), True),                          # invalid root actions (incl. a row with every action invalid)
    (40, 24, 0.5, dict(use_mixed=False, max_considered=1), True),
    (40, 24, 0.0, dict(max_considered=5), True),
    (17, 8, 0.0, dict(two_player=True), False),                        # negated discount (context.py:143)
    (50, 24, 2.0, dict(bonus_after_rescale=True), False),
    (50, 24, 1.0, dict(var_backup=1, max_depth=9), False),
])
def test_subleq16_search_run_whole_tree(oracle, eaz, torch, kernel_variant, B, S, beta_max, opts, masked):
    """The emctx call shape on Subleq-16 (begin -> run -> finish, the path of evaluate / reanalyze / a caller's own loop): every tree
    field, the state blobs of visited nodes, action, weights and summary against the oracle.  Under the default variant an eligible
    search is ONE launch of subleq16_selfplay_kernel in its search-only mode (root read from, whole tree written back to the mctx
    arrays) -- asserted through eaz_search_launch_info and the launch counters; S = 69 (trees beyond the 160 KiB of LDS) and the other variants
    take their own kernels and must agree all the same."""
    from e_alphazero_amd import engine
    from helpers import subleq_setup, subleq_device_to_oracle

    W = 16
    dims, params, state = subleq_setup(eaz, W, seed=B + S, fill=0.4)
    env = engine.EnvHandle.subleq(W, (1, 3), True, device=0)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    oenv = oracle.OracleEnv.subleq(W, (1, 3), True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"], hash_len=dims["hash_len"],
                            bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    # root states: a few random steps into the episodes, so that programs, step counts and reachable rewards differ across the batch
    st = env.init(B, key=7)
    rng = np.random.default_rng(B * 131 + S)
    for t in range(5):
        a = rng.integers(0, W, B).astype(np.int32)
        a[rng.random(B) < 0.5] = 0 if t else W - 2
        st = env.step(st, torch.from_numpy(a).cuda(), key=50 + t, autoreset=False)
    blobs = subleq_device_to_oracle(st.cpu().numpy(), W)
    ex, xp, v, u, _ = onet.forward(oenv.observe(blobs))
    beta = (beta_max * np.linspace(0, 1, B)).astype(np.float32)
    g = oracle.gumbel((61, 62), B * W).reshape(B, W)
    invalid = None
    if masked:
        invalid = (rng.random((B, W)) < 0.4).astype(np.uint8)
        invalid[0] = 1
        invalid[1] = 1
        invalid[1, 9] = 0
    explore = beta_max > 0
    logits = xp if explore else ex
    okw = dict(discount=0.97, use_exploration_logits=explore, rescale=opts.get("rescale", True), use_mixed=opts.get("use_mixed", True),
               max_considered=opts.get("max_considered", 16), max_depth=opts.get("max_depth", 0), two_player=opts.get("two_player", False),
               bonus_after_rescale=opts.get("bonus_after_rescale", False), var_backup=opts.get("var_backup", 0))
    otree = oracle.OracleTree(oenv, onet, B, S)
    otree.begin(logits, v, u, blobs, beta, invalid, g, **okw)
    for sim in range(S):
        otree.sim(sim)
    search = engine.Search(env, net, B, S, discount=0.97, exploration=explore, rescale_values=okw["rescale"], use_mixed_value=okw["use_mixed"],
                           max_num_considered_actions=okw["max_considered"], max_depth=okw["max_depth"] or None,
                           two_players_game=okw["two_player"], epistemic_bonus_after_rescale=okw["bonus_after_rescale"],
                           epistemic_variance_backup=okw["var_backup"])
    cu = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for rep in range(2):  # the second search reuses the (dirty) tree buffers of the first
        search.begin(cu(logits), cu(v), cu(u), st, cu(beta), cu(invalid), cu(g))
        search.profile(True)
        search.profile_read()
        search.run()
        torch.cuda.synchronize()
        prof = search.profile_read()
        search.profile(False)
        info = search.launch_info()
        assert info["one_launch"] == (kernel_variant == "persist" and S <= 68), (info, S)
        assert info["lds_bytes"] <= 160 * 1024
        if info["one_launch"]:
            assert prof["search"]["launches"] == 1 and prof["tree"]["launches"] == 0 and prof["net"]["launches"] == 0, prof
        act, aw = search.finish()
        oact, oaw = otree.finish()
        exact = _compare_tree(search, otree, f"run rep={rep}")
        emb = subleq_device_to_oracle(search.field("embeddings").cpu().numpy().reshape(B * (S + 1), -1), W).reshape(B, S + 1, -1)
        visited = search.dump()["node_visits"] > 0
        assert np.array_equal(emb[visited], otree.field("embeddings")[visited]), "state blobs of visited nodes differ"
        assert np.array_equal(act.cpu().numpy(), oact)
        np.testing.assert_allclose(aw.cpu().numpy(), oaw, rtol=0, atol=ATOL)
        s_, os_ = search.summary(), otree.summary()
        for k in os_:
            np.testing.assert_allclose(s_[k].cpu().numpy(), os_[k], rtol=0, atol=ATOL, err_msg=k)
    if opts.get("max_depth"):
        assert (search.dump()["node_visits"][:, 1:] > 1).any(), "the depth limit never re-expanded a node"
    print(f"Subleq-16 run(): tree float fields bit-exact vs oracle: {exact}")


def test_search_begin_then_stepwise_after_persist_default(oracle, eaz, torch):
    """With the persistent launch enabled eaz_search_begin skips the tree clear; stepping the same search simulation by simulation
    (eaz_search_simulate) must then clear lazily without losing the root, also on buffers a previous run() left dirty."""
    G, B, S = 8, 37, 12
    am, params, state = deepsea_setup(eaz, G, seed=21, fill=0.4)
    engine, env, net = _engine(eaz, G, am, params, state)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    blobs = random_states_deepsea(G, B, seed=22)
    ex, xp, v, u, _ = onet.forward(oenv.observe(blobs))
    beta = np.zeros(B, np.float32)
    g = oracle.gumbel((41, 42), B * 2).reshape(B, 2)
    otree = oracle.OracleTree(oenv, onet, B, S)
    otree.begin(ex, v, u, blobs, beta, None, g, discount=0.997)
    for sim in range(S):
        otree.sim(sim)
    search = engine.Search(env, net, B, S, discount=0.997)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (cu(ex), cu(v), cu(u), cu(blobs), cu(beta), None, cu(g))
    search.begin(*args)
    search.run()
    search.begin(*args)
    for sim in range(S):
        search.simulate(sim)
    act, _ = search.finish()
    _compare_tree(search, otree, "stepwise after run")
    assert np.array_equal(act.cpu().numpy(), otree.finish()[0])
    search.begin(*args)      # begin -> finish without a single simulation: the root-only tree
    act0, aw0 = search.finish()
    otree.begin(ex, v, u, blobs, beta, None, g, discount=0.997)
    oact0, oaw0 = otree.finish()
    _compare_tree(search, otree, "root only")
    assert np.array_equal(act0.cpu().numpy(), oact0)
    s0, os0 = search.summary(), otree.summary()
    assert np.array_equal(s0["visit_probs"].cpu().numpy(), os0["visit_probs"]) and not os0["visit_probs"].any()  # counts / max(total, 1)


@pytest.mark.parametrize("G,B,S,T,flags", [
    (8, 64, 16, 10, {}),
    (8, 64, 16, 6, dict(directed_exploration=True, exploration_beta=1.0)),
    (8, 48, 16, 5, dict(sample_actions=True)),
    (8, 48, 16, 5, dict(sample_from_improved_policy=True, uniform_search_policy=True)),
    (32, 256, 64, 3, {}),
    (8, 1, 8, 9, {}),      # a single environment
    (6, 17, 8, 6, {}),     # ragged: B not a multiple of any tile, observation size 36 (not a multiple of 16: byte staging path)
    (66, 20, 12, 3, {}),   # a grid whose action map (4 356 bytes) does not fit the one-launch kernel's LDS copy: read from global memory
])
def test_selfplay_matches_oracle(oracle, eaz, torch, G, B, S, T, flags):
    am, params, state = deepsea_setup(eaz, G, seed=5, fill=0.25)
    engine, env, net = _engine(eaz, G, am, params, state)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    search = engine.Search(env, net, B, S, exploration=flags.get("directed_exploration", False))
    st = env.init(B, key=1)
    ost = oenv.init(B)
    st_eager = st.clone()
    out = search.selfplay(st, T, key=99, **flags)  # production path: one captured hipGraph per real step, replayed T times
    out_eager = search.selfplay(st_eager, T, key=99, use_graph=False, **flags)  # plain stream launches
    assert all(torch.equal(out[k], out_eager[k]) for k in out) and torch.equal(st, st_eager)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(99, 0), **flags)
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(out["state"].cpu().numpy(), ref["state"])
    assert np.array_equal(st.cpu().numpy(), ost)
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)
