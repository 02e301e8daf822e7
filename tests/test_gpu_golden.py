"""GPU parity against the committed golden vectors (tests/golden/*.npz) and Subleq parity against the oracle."""
import os
import sys

import numpy as np
import pytest

from helpers import subleq_device_to_oracle, subleq_oracle_to_device, subleq_setup, deepsea_setup, random_states_deepsea, oracle_deepsea

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("kernel_variant")]
ATOL = 1e-5
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available()
    return t


# The oracle's answer for a test case does not depend on the kernel variant the device runs (tests/conftest.py): computed once per case and
# reused by the other variants -- the CPU oracle is most of this file's run time (and the GPU box of the driver has fewer host cores to spare).
_ORACLE_CACHE = {}


def _cached(key, fn):
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]


def _build_engine(eaz, kind, size, seed, fill, constants_first=False):
    from e_alphazero_amd import engine

    if kind == "deepsea":
        am, params, state = deepsea_setup(eaz, size, seed=seed, fill=fill)
        env = engine.EnvHandle.deepsea(size, am, device=0)
        net = engine.FCNet(env.ctx, size * size, 256, 2, params, state)
        return engine, env, net, 0.997
    if kind == "connect4":
        params, state = eaz.params.init_resnet_params(6, 7, 2, 7, 64, size, seed=seed)
        state[eaz.params.RESNET_HASH_MODULE] = {"binary_set": eaz.params.init_hash_state(fill=fill, seed=seed)[eaz.params.HASH_MODULE]["binary_set"]}
        env = engine.EnvHandle.connect4(device=0)
        return engine, env, engine.ResNet(env.ctx, 6, 7, 2, 7, params, state, num_blocks=size), 0.997
    dims, params, state = subleq_setup(eaz, size, seed=seed, fill=fill)
    env = engine.EnvHandle.subleq(size, (1,), True, device=0)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, size, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"],
                       input_order=eaz.params.subleq_constants_first(size, True) if constants_first else None)
    return engine, env, net, 0.97


def _to_oracle_states(kind, size, dev_states):
    a = dev_states.cpu().numpy()
    return subleq_device_to_oracle(a, size) if kind == "subleq" else a


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_selfplay_and_tree_match_golden(eaz, torch, name):
    kind, size, B, S, T, seed, fill, flags = MG.CASES[name]
    z = np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    flags = dict(flags)
    engine, env, net, discount = _build_engine(eaz, kind, size, seed, fill, flags.pop("constants_first", False))
    explore = bool(flags.get("directed_exploration", False))
    two_player = bool(flags.pop("two_player", False))
    search = engine.Search(env, net, B, S, discount=discount, exploration=explore, two_players_game=two_player)
    st = env.init(B, key=seed)
    out = search.selfplay(st, T, key=1000 + seed, **flags)
    assert np.array_equal(out["action"].cpu().numpy(), z["sp_action"])
    assert np.array_equal(_to_oracle_states(kind, size, out["state"]), z["sp_state"])
    assert np.array_equal(_to_oracle_states(kind, size, st), z["last_states"])
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), z[f"sp_{k}"], rtol=0, atol=ATOL, err_msg=k)
    # standalone search from the final states with the golden root outputs / noise injected: full tree comparison
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    search.begin(cu(z["root_logits"]), cu(z["root_value"]), cu(z["root_var"]), st, cu(z["beta"]), None, cu(z["gumbel"]))
    search.run()
    act, aw = search.finish()
    assert np.array_equal(act.cpu().numpy(), z["search_action"])
    np.testing.assert_allclose(aw.cpu().numpy(), z["search_action_weights"], rtol=0, atol=ATOL)
    d = search.dump()
    for k in MG.TREE_INT:
        assert np.array_equal(d[k], z[f"tree_{k}"]), k
    for k in MG.TREE_F32:
        np.testing.assert_allclose(d[k], z[f"tree_{k}"], rtol=0, atol=ATOL, err_msg=k)
    s = search.summary()
    for k in ("value", "value_epistemic_std", "qvalues", "qvalues_epistemic_variance", "visit_counts", "visit_probs"):
        np.testing.assert_allclose(s[k].cpu().numpy(), z[f"sum_{k}"], rtol=0, atol=ATOL, err_msg=k)


@pytest.mark.parametrize("W,binary,tasks", [(16, True, (1,)), (16, False, (3,)), (32, True, (2, 5, 9)), (256, True, (1,))])
def test_subleq_env_matches_oracle(oracle, eaz, torch, W, binary, tasks):
    from e_alphazero_amd import engine

    env = engine.EnvHandle.subleq(W, tasks, binary, device=0)
    oenv = oracle.OracleEnv.subleq(W, tasks, binary)
    assert env.obs_shape == oenv.obs_shape and env.num_actions == W
    B = 130
    st = env.init(B, key=5)
    ost = oenv.init(B, key=(5, 0))
    assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost)
    rng = np.random.default_rng(W)
    for t in range(W - 1 if W <= 32 else 14):
        a = rng.integers(0, W, B).astype(np.int32)
        a[: B // 4] = [W - 2, W - 3][t] if t < 2 else a[: B // 4]  # a quarter of the batch solves NEGATION_POSITIVE-style tasks
        auto = t % 3 == 2
        st = env.step(st, torch.from_numpy(a).cuda(), key=100 + t, autoreset=auto)
        ost = oenv.step(ost, a, key=(100 + t, 0), autoreset=auto)
        assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost), f"t={t}"
        assert np.array_equal(env.observe(st).cpu().numpy(), oenv.observe(ost)), f"obs t={t}"
        f, of = env.fields(st), oenv.fields(ost)
        assert np.array_equal(f["rewards"].cpu().numpy()[:, 0], of["rewards"])
        assert np.array_equal(f["terminated"].cpu().numpy(), of["terminated"])
    # round trip of the layout converters themselves
    assert np.array_equal(subleq_oracle_to_device(ost), st.cpu().numpy())


@pytest.mark.parametrize("W,tasks,B", [(16, (1,), 8192), (16, (2, 4, 6), 4096), (32, (1,), 2048), (256, (5,), 512)])
def test_subleq_interpreter_on_random_programs(oracle, eaz, torch, W, tasks, B):
    """Stress for the interpreter's exact shortcuts (period skip, IO-free fast path, sequential test cases with early out, the
    register-resident memory of word_size 16): thousands of uniformly random programs -- tight loops, counting loops that mutate
    memory for dozens of cycles, input readers, output writers, out-of-range cursors -- must land in exactly the state the
    literal 200-cycle oracle loop reaches."""
    from e_alphazero_amd import engine
    from helpers import ORACLE_SUBLEQ

    env = engine.EnvHandle.subleq(W, tasks, True, device=0)
    oenv = oracle.OracleEnv.subleq(W, tasks, True)
    rng = np.random.default_rng(100 + W)
    ost = oenv.init(B, key=(9, 0))
    o = ost.view(ORACLE_SUBLEQ).reshape(B)
    nwords = rng.integers(0, W - 3, B)                         # how many words of the program are already written
    mem = rng.integers(0, W, (B, 256)).astype(np.int32)
    # bias some programs towards the IO addresses so that readers / writers / halting programs are common
    io = rng.random((B, 256)) < 0.15
    mem[io] = rng.integers(W - 4, W, int(io.sum()))
    mem[np.arange(256)[None, :] >= nwords[:, None]] = 0
    o["memory"] = mem
    o["step_count"] = nwords
    ost = o.view(np.uint8).reshape(B, -1)
    st = torch.from_numpy(subleq_oracle_to_device(ost)).cuda()
    for t in range(3):
        a = rng.integers(0, W, B).astype(np.int32)
        st = env.step(st, torch.from_numpy(a).cuda(), key=300 + t)
        ost = oenv.step(ost, a, key=(300 + t, 0))
        assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost), f"W={W} t={t}"
        assert np.array_equal(env.observe(st).cpu().numpy(), oenv.observe(ost))
    solved = ost.view(ORACLE_SUBLEQ).reshape(B)["solved"]
    print(f"W={W}: {int(solved.sum())} of {B} random programs solved their task")


@pytest.mark.parametrize("lazy_rows", [0, 2], ids=["cleared-rows", "lazy-rows"])
@pytest.mark.parametrize("W,B,S", [(32, 12, 12), (64, 10, 10), (64, 5, 48), (128, 9, 8), (256, 8, 8), (256, 3, 40)])
def test_wide_action_sets_selfplay_matches_oracle(oracle, eaz, torch, W, B, S, lazy_rows):
    """Every lanes-per-tree / actions-per-lane instantiation of the tree kernels (32x1, 64x1, 64x2, 64x4) with the matching FC
    output widths (VALU fallback, 256-column layer 3) and LDS-resident Subleq memory: Subleq word_size 32 .. 256 self-play (64 .. 256:
    tree step + leaf step in one launch, subleq_tree_leaf_kernel<1 / 2 / 4>).  Each with the [B,N,A] tree arrays cleared per real step and
    with lazy child rows (EAZ_TUNE_LAZY_ROWS: the arrays are never cleared, TreeDev::lazy_rows); the S = 40 / 48 cases revisit edges of
    deep trees many times."""
    from e_alphazero_amd import engine

    engine.tune(lazy_rows=lazy_rows)
    dims, params, state = subleq_setup(eaz, W, seed=W + 1, fill=0.4)
    env = engine.EnvHandle.subleq(W, (1, 3), True, device=0)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    oenv = oracle.OracleEnv.subleq(W, (1, 3), True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    search = engine.Search(env, net, B, S, discount=0.97)
    st = env.init(B, key=7)
    ost = oenv.init(B, key=(7, 0))
    assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost)
    T = 3
    try:
        if lazy_rows:  # what the arrays held before must not matter: fill every [B,N,A] array with 0xFF bytes first (NaN, index / visits -1)
            search.debug_poison(0xFF, child_rows=True)
        out = search.selfplay(st, T, key=41)
    finally:
        engine.tune(lazy_rows=1)
    ref = _cached(("wide_selfplay", W, B, S), lambda: oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(41, 0), discount=0.97))
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(subleq_device_to_oracle(out["state"].cpu().numpy(), W), ref["state"])
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)


@pytest.mark.parametrize("flags", [{}, dict(directed_exploration=True, exploration_beta=1.0), dict(sample_actions=True),
                                   dict(sample_from_improved_policy=True, uniform_search_policy=True),
                                   dict(full_obs_hash=True), dict(full_obs_hash=True, constants_first=True, directed_exploration=True, exploration_beta=1.0)],
                         ids=["default", "beta1", "sample-visits", "sample-improved-uniform", "full-obs-hash", "full-obs-hash-constants-first-beta1"])
def test_subleq16_selfplay_crosses_episode_boundaries(oracle, eaz, torch, flags, kernel_variant):
    """Subleq-16 self-play over more real steps than an episode has (episodes end at _step_count >= 13, subleq.py:671-673): searches from
    terminated roots, the auto-reset inside selfplay (selfplay.py:26-75,135) with a task drawn per reset (subleq.py:624), a ragged batch
    (40 = 2.5 workgroups of the one-launch kernel), every action-choice flag.  Generated Gumbel noise (the engine's key derivation)."""
    from e_alphazero_amd import engine

    flags = dict(flags)
    full_obs_hash, constants_first = flags.pop("full_obs_hash", False), flags.pop("constants_first", False)
    if kernel_variant in ("tile16-unfused", "tile64-sharded") and flags and not full_obs_hash:
        pytest.skip("the per-simulation variants run this window once (default flags); the golden c3r / c3b cases cover them at T = 16 as well")
    W, B, S, T = 16, 40, 24, 17
    tasks = (1, 2, 3)
    # full_obs_hash: subleq_hash_only_io = False (config.py:30, fully_connected.py:84-89), the reference's "E-AZ (full-obs hash)" curve
    dims, params, state = subleq_setup(eaz, W, seed=11, fill=0.4, hash_io=not full_obs_hash)
    order = eaz.params.subleq_constants_first(W, True) if constants_first else None
    env = engine.EnvHandle.subleq(W, tasks, True, device=0)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"], input_order=order)
    oenv = oracle.OracleEnv.subleq(W, tasks, True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    onet.set_input_order(order)
    explore = bool(flags.get("directed_exploration", False))
    search = engine.Search(env, net, B, S, discount=0.97, exploration=explore)
    assert search.launch_info()["one_launch"] == (kernel_variant == "persist"), "Subleq-16 self-play runs as ONE launch for both hash slices and both layer-1 input orders"
    st = env.init(B, key=13)
    ost = oenv.init(B, key=(13, 0))
    assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost)
    out = search.selfplay(st, T, key=57, **flags)
    ost_in = ost.copy()
    ref, ost = _cached(("sq16_crossing", tuple(sorted(flags.items())), full_obs_hash, constants_first),
                       lambda: (oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(57, 0), discount=0.97, **flags), ost))
    assert ost_in.shape == ost.shape
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(subleq_device_to_oracle(out["state"].cpu().numpy(), W), ref["state"])
    assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost)
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)
    sc = ref["state"].view(np.uint8)[..., 0:4].copy().view(np.int32)[..., 0]
    assert (np.diff(sc.astype(np.int64), axis=0) < 0).any(), "no episode ended inside the window: the test lost its point"


@pytest.mark.parametrize("W,binary", [(17, True), (18, True), (30, False)])
def test_subleq_word_sizes_that_are_not_multiples_of_four(oracle, eaz, torch, W, binary, kernel_variant):
    """subleq.py:606 allows every word_size in 16..256; with subleq_hash_only_io (config.py:30) the hashed slice is 32 rows and always a
    multiple of 4 (hashes.py:210).  Environment steps on random programs, then self-play across an auto-reset, against the oracle."""
    from e_alphazero_amd import engine

    if kernel_variant not in ("persist", "tile16-unfused", "perlane-unfused"):
        pytest.skip("one fused and one unfused variant of the wavefront-per-environment kernel, and the lane-per-environment kernel (perlane-unfused)")
    B, S, T = 21, 10, 3
    dims, params, state = subleq_setup(eaz, W, binary=binary, seed=W, fill=0.3)
    assert dims["hash_len"] % 4 == 0 and W % 4 != 0
    env = engine.EnvHandle.subleq(W, (1, 4), binary, device=0)
    oenv = oracle.OracleEnv.subleq(W, (1, 4), binary)
    rng = np.random.default_rng(W)
    st = env.init(B, key=3)
    ost = oenv.init(B, key=(3, 0))
    for t in range(W + 2):  # past the end of an episode: finish, terminal no-op, auto-reset
        a = rng.integers(0, W, B).astype(np.int32)
        st = env.step(st, torch.from_numpy(a).cuda(), key=50 + t, autoreset=t % 3 == 2)
        ost = oenv.step(ost, a, key=(50 + t, 0), autoreset=t % 3 == 2)
        assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost), f"W={W} t={t}"
        assert np.array_equal(env.observe(st).cpu().numpy(), oenv.observe(ost))
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    search = engine.Search(env, net, B, S, discount=0.97)
    out = search.selfplay(st, T, key=61)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(61, 0), discount=0.97)
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(subleq_device_to_oracle(out["state"].cpu().numpy(), W), ref["state"])
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)


@pytest.mark.parametrize("case", [
    dict(B=1, S=8, T=3),                                                        # a single environment: 15 of the workgroup's trees mirror it
    dict(B=17, S=68, T=2),                                                      # the largest trees whose workgroup set fits into LDS (69 nodes)
    dict(B=17, S=69, T=2),                                                      # one more: the call falls back to the per-search kernels
    dict(B=33, S=16, T=15, rescale=False, gumbel_scale=0.5, max_considered=4),  # qtransform / root-selection options
    dict(B=20, S=12, T=14, two_player=True),                                    # negated discount (context.py:143) in the node records
    dict(B=24, S=16, T=15, reward_fn="lowest_bytes", explicit_gumbel=True, tasks=(1, 5)),
    dict(B=18, S=1, T=14),                                                      # the smallest searches: one / two simulations per real step
    dict(B=18, S=2, T=14, max_considered=2),
], ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k not in ("tasks",)))
def test_subleq16_one_launch_selfplay_edges(oracle, eaz, torch, case, kernel_variant):
    """Edges of subleq16_selfplay_kernel's eligibility and options (the other variants run the same inputs through their own kernels): batch
    sizes that leave mirror trees in a workgroup, the LDS limit on the tree size and the first size beyond it, the search options self-play
    exposes, the two-player discount, the lowest_bytes reward with a caller's root noise -- all across at least one episode boundary
    where T allows."""
    from e_alphazero_amd import engine

    if kernel_variant not in ("persist", "persist-step", "auto-fused"):
        pytest.skip("one-launch variants + one per-simulation variant")
    c = dict(case)
    W, B, S, T = 16, c.pop("B"), c.pop("S"), c.pop("T")
    tasks, reward_fn, explicit = c.pop("tasks", (1,)), c.pop("reward_fn", "solved_or_not"), c.pop("explicit_gumbel", False)
    dims, params, state = subleq_setup(eaz, W, seed=B + S, fill=0.4)
    env = engine.EnvHandle.subleq(W, tasks, True, device=0, reward_fn=reward_fn)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    oenv = oracle.OracleEnv.subleq(W, tasks, True, reward_fn=reward_fn)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    search = engine.Search(env, net, B, S, discount=0.97, two_players_game=c.get("two_player", False), rescale_values=c.get("rescale", True),
                           gumbel_scale=c.get("gumbel_scale", 1.0), max_num_considered_actions=c.get("max_considered", 16))
    if kernel_variant == "persist":
        assert search.launch_info()["one_launch"] == (S <= 68), (search.launch_info(), S)  # the LDS limit is where the docs say it is
    st = env.init(B, key=19)
    ost = oenv.init(B, key=(19, 0))
    gum = oracle.gumbel((8, 1), T * B * W).reshape(T, B, W) if explicit else None
    out = search.selfplay(st, T, key=73, gumbel=None if gum is None else torch.from_numpy(gum).cuda())
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(73, 0), discount=0.97, gumbel=gum, **c)
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(subleq_device_to_oracle(out["state"].cpu().numpy(), W), ref["state"])
    assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost)
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)


def test_deepsea_states_beyond_the_packed_form_take_the_per_simulation_path(oracle, eaz, torch, kernel_variant):
    """The one-launch DeepSea kernel keeps a state as one word (step count and column in 12 bits each, two flag bits, "reward is 1.0").
    Everything the environment produces fits; a caller's hand-made blob may not.  eaz_selfplay_run / eaz_search_begin / eaz_evaluate_run check
    the blobs a call starts from (ds_roots_fit_kernel) and send a batch with a misfit to the per-simulation kernels instead of truncating it:
    every variant, the default one included, gives the oracle's result.  (A column off the grid or a negative step count is outside the
    contract -- the reference would index with it -- but stays memory-safe: second half of the test.)"""
    from e_alphazero_amd import engine

    G, B, S, T = 8, 20, 12, 3
    am, params, state = deepsea_setup(eaz, G, seed=31, fill=0.4)
    env = engine.EnvHandle.deepsea(G, am, device=0)
    net = engine.FCNet(env.ctx, G * G, 256, 2, params, state)
    oenv, onet = oracle_deepsea(oracle, eaz, G, am, params, state)
    blobs = random_states_deepsea(G, B, seed=32)
    i32 = blobs.view(np.int32).reshape(B, 4)
    f32 = blobs.view(np.float32).reshape(B, 4)
    i32[2, 0] = 5000        # a step count beyond 12 bits (the row read is clamped, the episode is over)
    f32[9, 2] = 0.5         # a reward that is neither 0 nor 1
    blobs[11, 12] = 7       # `terminated` as a byte other than 0 / 1
    search = engine.Search(env, net, B, S, discount=0.997)
    if kernel_variant in ("persist", "persist-step"):
        assert search.launch_info()["one_launch"], "eligible by shape: only the blobs' content routes this call"
    st = torch.from_numpy(blobs.copy()).cuda()
    ost = blobs.copy()
    out = search.selfplay(st, T, key=83)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(83, 0), discount=0.997)
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(out["state"].cpu().numpy(), ref["state"])
    assert np.array_equal(st.cpu().numpy(), ost)
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)
    # the same blobs through evaluate (greedy search, plain env.step until every environment has terminated)
    ev = engine.Search(env, net, B, S, discount=0.997, gumbel_scale=0.0)
    st2, ost2 = torch.from_numpy(blobs.copy()).cuda(), blobs.copy()
    ret, iters = ev.evaluate(st2, exploitation_beta=0.0, max_episode_length=G + 2)
    oret, oiters = oracle.evaluate(oenv, onet, ost2, S=S, exploitation_beta=0.0, max_episode_length=G + 2, discount=0.997)
    assert iters == oiters and np.array_equal(st2.cpu().numpy(), ost2)
    np.testing.assert_allclose(ret.cpu().numpy(), oret, rtol=0, atol=ATOL)
    # off the grid: no parity claim (jnp would wrap / clamp the indices its own way), but the call must come back, with finite numbers and
    # every environment on the grid after its first real step (the action-map read is clamped, the observation has no set cell)
    wild = blobs.copy()
    w32 = wild.view(np.int32).reshape(B, 4)
    w32[4, 0], w32[6, 1], w32[7, 1], w32[8, 1] = -3, G + 5, -2, 2 ** 31 - 1
    st3 = torch.from_numpy(wild).cuda()
    out3 = search.selfplay(st3, T, key=84)
    torch.cuda.synchronize()
    assert all(torch.isfinite(out3[k]).all() for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction"))
    pos3 = st3.cpu().numpy().view(np.int32).reshape(B, 4)[:, 1]
    assert ((pos3 >= 0) & (pos3 < G)).all()


def test_subleq16_states_beyond_the_compact_form_take_the_per_simulation_path(oracle, eaz, torch, kernel_variant):
    """The one-launch Subleq-16 kernel keeps a state as 32 bytes (4-bit program words, IO values as bytes with 16 = "beyond the valid ones",
    8-bit step count, 16-bit task).  Everything the environment produces fits; a caller's hand-made blob may not.  eaz_selfplay_run /
    eaz_search_begin check the blobs a call starts from (sq16_roots_fit_kernel) and send a batch with a misfit to the per-simulation
    kernels instead of truncating it: every variant, the default one included, gives the oracle's result."""
    from e_alphazero_amd import engine
    from helpers import DEVICE_SUBLEQ

    W, B, S, T = 16, 20, 12, 3
    dims, params, state = subleq_setup(eaz, W, seed=5, fill=0.4)
    env = engine.EnvHandle.subleq(W, (1, 3), True, device=0)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    oenv = oracle.OracleEnv.subleq(W, (1, 3), True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    ost, _ = oracle.random_play(oenv, B, 5, key=(21, 0))               # mid-episode states
    d = subleq_oracle_to_device(ost).view(DEVICE_SUBLEQ).reshape(B).copy()
    d["ex_in_after"][4, 2] = 40                                        # an IO value the byte-with-16-as-padding form cannot hold
    d["task"][9] = 70000                                               # beyond 16 bits (get_test_cases clamps it to the last branch)
    d["ex_out"][11, 5] = 300
    ost = subleq_device_to_oracle(d.view(np.uint8).reshape(B, 336), W)
    ost0 = ost.copy()                                                  # selfplay advances its states in place
    st = torch.from_numpy(d.view(np.uint8).reshape(B, 336).copy()).cuda()
    search = engine.Search(env, net, B, S, discount=0.97)
    out = search.selfplay(st, T, key=91)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(91, 0), discount=0.97)
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(subleq_device_to_oracle(out["state"].cpu().numpy(), W), ref["state"])
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)
    # the emctx call shape from the same roots: begin -> run -> finish
    ex, xp, v, u, _ = onet.forward(oenv.observe(ost0))
    g = oracle.gumbel((4, 4), B * W).reshape(B, W)
    beta = np.zeros(B, np.float32)
    act, aw, summ, tree = oracle.search(oenv, onet, ost0, S, ex, v, u, beta, g, discount=0.97)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    st2 = torch.from_numpy(d.view(np.uint8).reshape(B, 336).copy()).cuda()
    search.begin(cu(ex), cu(v), cu(u), st2, cu(beta), None, cu(g))
    search.run()
    a2, aw2 = search.finish()
    assert np.array_equal(a2.cpu().numpy(), act)
    np.testing.assert_allclose(aw2.cpu().numpy(), aw, rtol=0, atol=ATOL)
    dd, od = search.dump(), tree.dump()
    for k in MG.TREE_INT:
        assert np.array_equal(dd[k], od[k]), k


@pytest.mark.parametrize("env_kind", ["subleq", "deepsea"])
def test_poisoned_select_arrays_cost_neither_time_nor_results(oracle, eaz, torch, kernel_variant, env_kind):
    """Regression for round 3's 257-second launch: in a one-launch search the lane groups beyond a ragged batch mirrored
    tree 0 of ANOTHER workgroup and read its path length before it was written; a leftover 0x7F7F7F7F from a freed allocation made the
    backward loop run 2^28 times (results correct, one launch of 257 s).  Here the select scratch is filled with exactly that pattern
    between begin() and run() (eaz_search_debug_poison), on a ragged batch, for every kernel variant: the search has to finish in its
    normal time AND agree with the oracle -- a depth clamp that merely masked a stale read would pass the second but not a future
    regression of the first, so both are asserted."""
    import time

    from e_alphazero_amd import engine

    if env_kind == "subleq":
        W, B, S = 16, 37, 24   # 2 workgroups of 16 trees + 5: the last workgroup of the one-launch kernels has 11 mirror groups
        dims, params, state = subleq_setup(eaz, W, seed=3, fill=0.4)
        env = engine.EnvHandle.subleq(W, (1,), True, device=0)
        net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
        oenv = oracle.OracleEnv.subleq(W, (1,), True)
        onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                                hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
        A, to_dev = W, lambda o: torch.from_numpy(subleq_oracle_to_device(o)).cuda()
    else:
        G, B, S = 8, 37, 24
        am, params, state = deepsea_setup(eaz, G, seed=3, fill=0.4)
        env = engine.EnvHandle.deepsea(G, am, device=0)
        net = engine.FCNet(env.ctx, G * G, 256, 2, params, state)
        oenv = oracle.OracleEnv.deepsea(G, am)
        onet = oracle.OracleNet(G * G, 256, 2, eaz.params.fc_weight_list(params), bitset=state[eaz.params.HASH_MODULE]["binary_set"])
        A, to_dev = 2, lambda o: torch.from_numpy(o).cuda()
    ost, _ = oracle.random_play(oenv, B, 4, key=(31, 0))
    ex, xp, v, u, _ = onet.forward(oenv.observe(ost))
    g = oracle.gumbel((5, 5), B * A).reshape(B, A)
    beta = np.zeros(B, np.float32)
    act, aw, summ, tree = oracle.search(oenv, onet, ost, S, ex, v, u, beta, g, discount=0.97)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    search = engine.Search(env, net, B, S, discount=0.97)
    st = to_dev(ost)
    for rep in range(2):   # the second round runs on arrays the first one used
        search.begin(cu(ex), cu(v), cu(u), st, cu(beta), None, cu(g))
        search.debug_poison(0x7F)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        search.run()
        a2, aw2 = search.finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert dt < 5.0, f"the search took {dt:.1f} s: a stale select array is being trusted again"
        assert np.array_equal(a2.cpu().numpy(), act)
        np.testing.assert_allclose(aw2.cpu().numpy(), aw, rtol=0, atol=ATOL)
        dd, od = search.dump(), tree.dump()
        for k in MG.TREE_INT:
            assert np.array_equal(dd[k], od[k]), k


def test_empty_batches_are_no_ops(eaz, torch):
    """B = 0 through the C ABI: every batched entry point accepts an empty batch (null data pointers included)."""
    from e_alphazero_amd import engine

    env = engine.EnvHandle.deepsea(8, device=0)
    p = eaz.params.init_fc_params(64, 256, 2, seed=0)
    net = engine.FCNet(env.ctx, 64, 256, 2, p, eaz.params.init_hash_state())
    st = env.init(0)
    assert st.shape == (0, env.state_bytes)
    st = env.step(st, torch.zeros(0, dtype=torch.int32, device="cuda"))
    obs = env.observe(st).reshape(0, 64)
    assert obs.shape[0] == 0
    assert all(t.shape[0] == 0 for t in net.forward(obs))
    assert env.fields(st)["terminated"].shape == (0,)
    net.hash_update(obs)
    seen, idx = net.hash_lookup(obs)
    assert seen.shape == (0,)
    with pytest.raises(eaz.EazError):
        engine.Search(env, net, 0, 8)  # a search needs at least one tree


def test_random_play_matches_oracle(oracle, eaz, torch):
    """uniformrandomplay (selfplay.py:152-170): same keys -> same trajectories."""
    from e_alphazero_amd import engine

    am, _, _ = deepsea_setup(eaz, 8, seed=9)
    env = engine.EnvHandle.deepsea(8, am, device=0)
    oenv = oracle.OracleEnv.deepsea(8, am)
    last, traj = env.random_play(64, 20, key=77, want_traj=True)
    olast, otraj = oracle.random_play(oenv, 64, 20, key=(77, 0), want_traj=True)
    assert np.array_equal(last.cpu().numpy(), olast) and np.array_equal(traj.cpu().numpy(), otraj)
    senv = engine.EnvHandle.subleq(16, (1, 2, 3), True, device=0)
    osenv = oracle.OracleEnv.subleq(16, (1, 2, 3), True)
    last, _ = senv.random_play(96, 30, key=78)
    olast, _ = oracle.random_play(osenv, 96, 30, key=(78, 0))
    assert np.array_equal(subleq_device_to_oracle(last.cpu().numpy(), 16), olast)


def test_hash_update_matches_oracle(oracle, eaz, torch):
    """BaseHash.update (hashes.py:45-50) as a scatter-OR kernel."""
    G, B = 8, 500
    am, params, state = deepsea_setup(eaz, G, seed=1, fill=0.0)
    from e_alphazero_amd import engine

    env = engine.EnvHandle.deepsea(G, am, device=0)
    net = engine.FCNet(env.ctx, G * G, 256, 2, params, state)
    obs = (np.random.default_rng(0).random((B, G * G)) < 0.1).astype(np.uint8)
    net.hash_update(torch.from_numpy(obs).cuda())
    bs = net.get_state()[eaz.params.HASH_MODULE]["binary_set"]
    idx = np.empty(B, np.uint32)
    oracle.lib().eazo_xxhash_indices(obs.ctypes.data, B, G * G, 24, idx.ctypes.data)
    ref = np.zeros(1 << 21, np.uint8)
    oracle.lib().eazo_hash_update(ref.ctypes.data, idx.ctypes.data, B)
    assert np.array_equal(bs, ref)
    seen, _ = net.hash_lookup(torch.from_numpy(obs).cuda())
    assert seen.all()


@pytest.mark.parametrize("W,binary,B", [(16, True, 200), (16, False, 77), (256, True, 130), (32, False, 65)])
def test_fc_forward_subleq_shapes(oracle, eaz, torch, W, binary, B):
    """FC forward on Subleq observations: in_dim 240 / 544 / 2592 (multi-chunk staging) / 2112, out = 16 (MFMA layer 3 with LDS
    weights), 256 (layer-2 code path) and 32 (VALU fallback); hash_io slice."""
    from e_alphazero_amd import engine

    dims, params, state = subleq_setup(eaz, W, binary=binary, seed=W, fill=0.5)
    env = engine.EnvHandle.subleq(W, (1,), binary, device=0)
    oenv = oracle.OracleEnv.subleq(W, (1,), binary)
    assert env.obs_dim == dims["in_dim"]
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    ost, _ = oracle.random_play(oenv, B, 7, key=(W, 1))
    obs = oenv.observe(ost)
    ref = onet.forward(obs)
    out = net.forward(torch.from_numpy(obs).cuda())
    for n, a, b in zip(["exploit", "explore", "value", "ube", "novelty"], out, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=0, atol=ATOL, err_msg=n)
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(out, ref)), "FC forward is specified to be bit-exact"


@pytest.mark.parametrize("kind,size,B,S,beta", [("deepsea", 8, 64, 16, 0.0), ("subleq", 16, 48, 16, -0.5), ("subleq", 16, 1, 8, 0.0),
                                                ("subleq", 32, 3, 8, 0.0)])
def test_evaluate_matches_oracle(oracle, eaz, torch, kind, size, B, S, beta, kernel_variant):
    """N2: evaluate driver (src/evaluate.py:13-57): greedy search episodes, per-episode returns and loop trip count.  Subleq-16 under the
    default variant: the whole while_loop is ONE launch of the one-launch kernel in its evaluate mode (a device-side loop, like the
    reference's lax.while_loop) -- asserted through the launch counters: one search-class launch, no tree / network / env launches, i.e. no
    host round trip between the first and the last step."""
    engine, env, net, discount = _build_engine(eaz, kind, size, 7, 0.3)
    oenv, onet, _ = MG.build(kind, size, 7, 0.3)
    search = engine.Search(env, net, B, S, discount=discount, gumbel_scale=0.0)
    st = env.init(B, key=3)
    ost = oenv.init(B, key=(3, 0))
    max_len = 10 if kind == "subleq" else 1000  # config.py:123 max_episode_length for subleq
    search.profile(True)
    search.profile_read()
    ret, it = search.evaluate(st, exploitation_beta=beta, max_episode_length=max_len)
    torch.cuda.synchronize()
    prof = search.profile_read()
    search.profile(False)
    oret, oit = oracle.evaluate(oenv, onet, ost, S, discount=discount, exploitation_beta=beta, max_episode_length=max_len)
    assert it == oit and it >= (size - 1 if kind == "deepsea" else 1)
    assert np.array_equal(ret.cpu().numpy(), oret)
    assert np.array_equal(_to_oracle_states(kind, size, st), ost)
    if (kind == "deepsea" and kernel_variant in ("persist", "persist-step")) or (kind == "subleq" and size == 16 and kernel_variant == "persist"):
        assert prof["search"]["launches"] == 1 and prof["tree"]["launches"] == 0 and prof["net"]["launches"] == 0 and prof["gather"]["launches"] == 0, prof
    if kind in ("subleq", "deepsea"):
        # a second call from the (now terminal or capped) states: nothing left to do for terminated environments, trip count from the rest
        ret2, it2 = search.evaluate(st, exploitation_beta=beta, max_episode_length=max_len)
        oret2, oit2 = oracle.evaluate(oenv, onet, ost, S, discount=discount, exploitation_beta=beta, max_episode_length=max_len)
        assert it2 == oit2 and np.array_equal(ret2.cpu().numpy(), oret2) and np.array_equal(_to_oracle_states(kind, size, st), ost)
    if kind == "deepsea":
        # staggered starts: environments at different depths (and a cap below the episode length) end at different times per workgroup
        st3 = torch.from_numpy(random_states_deepsea(size, B, seed=5)).cuda()
        ost3 = st3.cpu().numpy().copy()
        for cap in (1000, 2):
            a, b = st3.clone(), ost3.copy()
            ret3, it3 = search.evaluate(a, exploitation_beta=beta, max_episode_length=cap)
            oret3, oit3 = oracle.evaluate(oenv, onet, b, S, discount=discount, exploitation_beta=beta, max_episode_length=cap)
            assert it3 == oit3 and np.array_equal(ret3.cpu().numpy(), oret3) and np.array_equal(a.cpu().numpy(), b), cap


def test_evaluate_from_staggered_subleq_states(oracle, eaz, torch, kernel_variant):
    """evaluate (src/evaluate.py:13-57) from states whose episodes end at different times -- one workgroup of the one-launch kernel is one
    step from the end, one needs ten, a ragged third is in between, two environments start terminal: every workgroup stops at its own trip
    count, the reference's loop runs until the last environment has terminated (the earlier ones take no-op steps that zero their rewards
    field: sq16_eval_finalize_kernel).  Returns, trip count and every byte of the final states against the oracle."""
    from e_alphazero_amd import engine

    W, B, S = 16, 40, 12
    dims, params, state = subleq_setup(eaz, W, seed=13, fill=0.4)
    env = engine.EnvHandle.subleq(W, (1, 2), True, device=0)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    oenv = oracle.OracleEnv.subleq(W, (1, 2), True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    ost = oenv.init(B, key=(41, 0))
    rng = np.random.default_rng(41)
    steps = np.r_[np.full(16, 12), np.full(16, 3), np.full(8, 7)]
    steps[[5, 20]] = 13   # two environments are already terminal (step_count >= W - 3, subleq.py:671-673)
    for t in range(13):
        a = rng.integers(0, W, B).astype(np.int32)
        nxt = oenv.step(ost.copy(), a, key=(t, 0))
        ost = np.where((t < steps)[:, None], nxt, ost)
    search = engine.Search(env, net, B, S, discount=0.97, gumbel_scale=0.0)
    st = torch.from_numpy(subleq_oracle_to_device(ost)).cuda()
    ret, it = search.evaluate(st, exploitation_beta=0.0, max_episode_length=1000)
    oret, oit = oracle.evaluate(oenv, onet, ost, S, discount=0.97, exploitation_beta=0.0, max_episode_length=1000)
    assert it == oit == 10, (it, oit)
    assert np.array_equal(ret.cpu().numpy(), oret)
    assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost)


@pytest.mark.parametrize("kind,size,B,S,kw", [
    ("deepsea", 8, 96, 16, dict(reanalyze_beta=0.0, exploration_beta=0.0)),
    ("deepsea", 8, 96, 16, dict(reanalyze_beta=-0.5, exploration_beta=1.0, exploration_ube_target=False, exploration_policy_target_temperature=2.0)),
    ("subleq", 16, 40, 32, dict(reanalyze_beta=-1.0, exploration_beta=1.0)),
])
def test_reanalyze_matches_oracle(oracle, eaz, torch, kind, size, B, S, kw):
    """N1: reanalyze target builder (src/reanalyze.py:52-131) on (state, next_state) pairs taken from a random-play trajectory."""
    engine, env, net, discount = _build_engine(eaz, kind, size, 9, 0.3)
    oenv, onet, _ = MG.build(kind, size, 9, 0.3)
    _, traj = oracle.random_play(oenv, B, 12, key=(5, 5), want_traj=True)
    rng = np.random.default_rng(0)
    t = rng.integers(0, 11, B)
    s0 = traj[t, np.arange(B)]
    s1 = traj[t + 1, np.arange(B)]  # contains terminal -> reset transitions as well (reanalyze.py:127 FIXME)
    g = oracle.gumbel((8, 8), B * env.num_actions).reshape(B, env.num_actions)
    ref = oracle.reanalyze(oenv, onet, s0, s1, S, g, discount=discount, **kw)
    conv = (lambda a: a) if kind == "deepsea" else __import__("helpers").subleq_oracle_to_device
    search = engine.Search(env, net, B, S, discount=discount)
    out = search.reanalyze(torch.from_numpy(conv(s0)).cuda(), torch.from_numpy(conv(s1)).cuda(), gumbel=torch.from_numpy(g).cuda(), **kw)
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    for k in ("value_target", "ube_target", "exploration_policy_target", "exploitation_policy_target"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)
    f = oenv.fields(s0)
    assert (ref["value_target"][f["terminated"]] == 0).all() and (ref["ube_target"][f["terminated"]] == 0).all()


# ---- C4: connect_four + EpistemicResidualAZNet ----------------------------------------------------------------------
def _c4_setup(oracle, eaz, nb, seed=0, fill=0.3):
    from e_alphazero_amd import engine

    params, state = eaz.params.init_resnet_params(6, 7, 2, 7, 64, nb, seed=seed)
    bs = eaz.params.init_hash_state(fill=fill, seed=seed)[eaz.params.HASH_MODULE]["binary_set"]
    state[eaz.params.RESNET_HASH_MODULE] = {"binary_set": bs}
    env = engine.EnvHandle.connect4(device=0)
    net = engine.ResNet(env.ctx, 6, 7, 2, 7, params, state, num_blocks=nb)
    oenv = oracle.OracleEnv.connect4()
    onet = oracle.OracleNet.resnet(6, 7, 2, 7, eaz.params.resnet_array_list(params, state, nb), 64, nb, bitset=bs)
    return engine, env, net, oenv, onet


def test_connect4_env_matches_oracle(oracle, eaz, torch):
    from e_alphazero_amd import engine

    env = engine.EnvHandle.connect4(device=0)
    oenv = oracle.OracleEnv.connect4()
    assert env.num_actions == 7 and env.num_players == 2 and env.obs_dim == 84 and env.state_bytes == 80
    B = 300
    st = env.init(B, key=11)
    ost = oenv.init(B, key=(11, 0))
    assert np.array_equal(st.cpu().numpy(), ost)
    rng = np.random.default_rng(0)
    for t in range(50):  # random play incl. illegal moves (full columns), wins, draws, terminal no-ops and auto-resets
        a = rng.integers(0, 7, B).astype(np.int32)
        auto = t % 4 == 3
        st = env.step(st, torch.from_numpy(a).cuda(), key=200 + t, autoreset=auto)
        ost = oenv.step(ost, a, key=(200 + t, 0), autoreset=auto)
        assert np.array_equal(st.cpu().numpy(), ost), f"t={t}"
        assert np.array_equal(env.observe(st).cpu().numpy(), oenv.observe(ost))
    f = env.fields(st)
    assert f["rewards"].shape == (B, 2) and f["legal_action_mask"].shape == (B, 7)
    assert f["terminated"].any() and not f["terminated"].all()


@pytest.mark.parametrize("nb,B", [(1, 50), (5, 97), (1, 3), (2, 51)])  # the last workgroup holds 2, 1, 3, 3 of its four samples
def test_resnet_forward_matches_oracle(oracle, eaz, torch, nb, B):
    engine, env, net, oenv, onet = _c4_setup(oracle, eaz, nb, seed=nb)
    ost, _ = oracle.random_play(oenv, B, 11, key=(1, 2))
    obs = oenv.observe(ost)
    ref = onet.forward(obs)
    out = net.forward(torch.from_numpy(obs).cuda())
    for n, a, b in zip(["exploit", "explore", "value", "ube", "novelty"], out, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=0, atol=ATOL, err_msg=n)
    print("resnet forward bit-exact vs oracle:", all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(out, ref)))


@pytest.mark.parametrize("H,W,cin,A,B", [(5, 5, 4, 20, 11), (3, 4, 2, 5, 6), (6, 7, 2, 16, 9)])
def test_resnet_forward_other_boards_and_action_counts(oracle, eaz, torch, H, W, cin, A, B):
    """The tower's kernel is specialised for 64 channels and at most 42 cells, not for connect_four: smaller boards (K of the heads' linear layers
    padded with zero operands), more input planes, and more actions than one MFMA tile has columns (the policy layers then run as plain chains)."""
    from e_alphazero_amd import engine

    nb = 2
    params, state = eaz.params.init_resnet_params(H, W, cin, A, 64, nb, seed=H * W + A)
    bs = eaz.params.init_hash_state(fill=0.3, seed=1)[eaz.params.HASH_MODULE]["binary_set"]
    state[eaz.params.RESNET_HASH_MODULE] = {"binary_set": bs}
    env = engine.EnvHandle.connect4(device=0)  # only its context is used
    net = engine.ResNet(env.ctx, H, W, cin, A, params, state, num_blocks=nb)
    onet = oracle.OracleNet.resnet(H, W, cin, A, eaz.params.resnet_array_list(params, state, nb), 64, nb, bitset=bs)
    obs = (np.random.default_rng(B).random((B, H * W * cin)) < 0.4).astype(np.uint8)
    ref = onet.forward(obs)
    out = net.forward(torch.from_numpy(obs).cuda())
    for n, a, b in zip(["exploit", "explore", "value", "ube", "novelty"], out, ref):
        assert np.array_equal(a.cpu().numpy(), b), n


@pytest.mark.parametrize("flags", [{}, dict(directed_exploration=True, exploration_beta=1.0)])
def test_connect4_selfplay_matches_oracle(oracle, eaz, torch, flags):
    """BASELINE.json configs[3] reduced: connect_four, two-player discount (-gamma), ResNet, sequential halving with m = 7."""
    engine, env, net, oenv, onet = _c4_setup(oracle, eaz, 2, seed=4)
    B, S, T = 40, 24, 12
    explore = bool(flags.get("directed_exploration", False))
    search = engine.Search(env, net, B, S, discount=0.997, two_players_game=True, exploration=explore)
    st = env.init(B, key=2)
    ost = oenv.init(B, key=(2, 0))
    out = search.selfplay(st, T, key=31, **flags)
    ref = _cached(("c4_selfplay", tuple(sorted(flags.items()))),
                  lambda: oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(31, 0), two_player=True, **flags))
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(out["state"].cpu().numpy(), ref["state"])
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)
    ret, it = engine.Search(env, net, B, S, discount=0.997, two_players_game=True, gumbel_scale=0.0).evaluate(env.init(B, key=5), max_episode_length=50)
    oret, oit = _cached(("c4_selfplay_eval",), lambda: oracle.evaluate(oenv, onet, oenv.init(B, key=(5, 0)), S, discount=0.997, two_player=True,
                                                                      max_episode_length=50))
    assert it == oit and np.array_equal(ret.cpu().numpy(), oret)


def test_connect4_evaluate_matches_oracle(oracle, eaz, torch):
    """N2 on the two-player configuration: evaluate (greedy episodes to the end of the game) on connect_four with the residual
    network; N1 (reanalyze) is single-player by construction in the reference."""
    engine, env, net, oenv, onet = _c4_setup(oracle, eaz, 1, seed=6)
    B, S = 24, 12
    search = engine.Search(env, net, B, S, discount=0.997, two_players_game=True, gumbel_scale=0.0)
    st = env.init(B, key=4)
    ost = oenv.init(B, key=(4, 0))
    ret, it = search.evaluate(st, exploitation_beta=0.0, max_episode_length=50)
    (oret, oit), ost = _cached(("c4_evaluate",), lambda: (oracle.evaluate(oenv, onet, ost, S, discount=0.997, two_player=True, exploitation_beta=0.0,
                                                                         max_episode_length=50), ost))
    assert it == oit and 7 <= it <= 42
    assert np.array_equal(ret.cpu().numpy(), oret)
    assert np.array_equal(st.cpu().numpy(), ost)
    # reanalyze is single-player in the reference (reanalyze.py:95 squeezes a [B,1] reward): the engine refuses the two-player env
    _, traj = oracle.random_play(oenv, B, 4, key=(6, 6), want_traj=True)
    g = torch.zeros(B, 7, device="cuda")
    with pytest.raises(eaz.EazError, match="single-player"):
        search.reanalyze(torch.from_numpy(traj[0]).cuda(), torch.from_numpy(traj[1]).cuda(), gumbel=g)


# ---- weight refresh: the facade must never search on stale weights (the reference re-applies model_params on every call) --------
@pytest.mark.parametrize("W,binary,B", [(16, True, 200), (32, True, 70), (256, True, 40), (16, False, 33)])
def test_fc_forward_follows_the_layer1_input_order(oracle, eaz, torch, W, binary, B):
    """The numerics switch eaz_net_set_input_order (include/eaz.h) against the oracle's eazo_net_set_input_order: every forward kernel
    (the variants of tests/conftest.py) computes the layer-1 chain in the configured order, bit for bit; a weight upload under an order
    lands in the right rows; n_seg = 0 restores the ascending default; the results of the two orders differ in the last bits only."""
    from e_alphazero_amd import engine

    dims, params, state = subleq_setup(eaz, W, binary=binary, seed=W + 3, fill=0.5)
    env = engine.EnvHandle.subleq(W, (1,), binary, device=0)
    oenv = oracle.OracleEnv.subleq(W, (1,), binary)
    order = eaz.params.subleq_constants_first(W, binary)
    net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"], input_order=order)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                            hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    ost, _ = oracle.random_play(oenv, B, 9, key=(W, 2))
    obs = oenv.observe(ost)
    dobs = torch.from_numpy(obs).cuda()
    ref_default = onet.forward(obs)
    onet.set_input_order(order)
    ref = onet.forward(obs)
    out = net.forward(dobs)
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(out, ref)), "the forward does not follow the configured input order"
    assert any(not np.array_equal(a, b) for a, b in zip(ref[:3], ref_default[:3])), "the order changed no bit: the test lost its point"
    for a, b in zip(ref, ref_default):
        np.testing.assert_allclose(a, b, rtol=0, atol=ATOL)
    # new weights under the order
    params2 = eaz.params.init_fc_params(dims["in_dim"], 256, W, seed=W + 4)
    net.set_params(params2)
    onet2 = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params2), hash_off=dims["hash_off"],
                             hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    onet2.set_input_order(order)
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(net.forward(dobs), onet2.forward(obs))), "weight upload under an input order"
    # back to the default
    net.set_input_order(None)
    onet2.set_input_order(None)
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(net.forward(dobs), onet2.forward(obs))), "restoring the ascending order"
    with pytest.raises(eaz.EazError, match="permutation"):
        net.set_input_order([(0, 10), (5, dims["in_dim"] - 5)])


def test_weights_mutated_in_place_reach_the_device_fc(oracle, eaz, torch):
    """api.selfplay with the SAME params dict whose arrays were overwritten in place (and with a replaced dict): the device copy is
    refreshed by content fingerprint, and the result equals the oracle on the new weights -- also through the captured step graph."""
    from e_alphazero_amd import api, engine

    G, B, S, T = 8, 48, 16, 5
    env = api.DeepSea(size_of_grid=G, action_map_key=api.random.PRNGKey(2))
    config = api.Config(selfplay_batch_size=B, selfplay_simulations_per_step=S, selfplay_steps=T)
    forward = api.get_forward_fn(env, config)
    model = forward.init(api.random.PRNGKey(3))
    context = api.Context(env=env, devices=[0], forward=forward, selfplay_recurrent_fn=api.get_epistemic_recurrent_fn(
        env=env, forward=forward, batch_size=B, exploration=False, discount=config.discount, two_players_game=False))
    states0 = api.vmap(env.init)(api.random.split(api.random.PRNGKey(5), B))

    def oracle_run(params, state):
        oenv = oracle.OracleEnv.deepsea(G, env.action_map)
        onet = oracle.OracleNet(G * G, 256, 2, eaz.params.fc_weight_list(params), bitset=state[eaz.params.HASH_MODULE]["binary_set"])
        ost = oenv.init(B)
        return oracle.selfplay(oenv, onet, ost, T=T, S=S, key=engine.key_pair(api.random.PRNGKey(11)), discount=config.discount)

    def check(tag):
        _, out = api.selfplay(model, config, context, states0, api.random.PRNGKey(11))
        ref = oracle_run(*model)
        assert np.array_equal(out.state.data.cpu().numpy(), ref["state"]), tag
        np.testing.assert_allclose(out.root_value.cpu().numpy(), ref["root_value"], rtol=0, atol=ATOL, err_msg=tag)
        np.testing.assert_allclose(out.ube_prediction.cpu().numpy(), ref["ube_prediction"], rtol=0, atol=ATOL, err_msg=tag)
        return out.root_value.clone()

    v0 = check("initial weights")
    rng = np.random.default_rng(9)
    for mod in model[0].values():                      # overwrite every array IN PLACE: same dicts, same ndarray objects
        for k in mod:
            mod[k][...] = (mod[k] + 0.05 * rng.standard_normal(mod[k].shape)).astype(np.float32)
    model[1][eaz.params.HASH_MODULE]["binary_set"][::7] ^= 0x5A   # and the novelty set
    v1 = check("weights mutated in place")
    assert not torch.equal(v0, v1), "the mutation did not change the search: the test does not cover what it claims"
    model = forward.init(api.random.PRNGKey(77))       # a brand-new pytree
    check("replaced pytree")


def test_resnet_weights_refresh_in_place_and_state_round_trip(oracle, eaz, torch):
    """eaz_net_set_resnet_weights repacks into the SAME device buffer: a search (and its captured graph) built before the update
    runs on the new weights; ResNet.get_state returns the model_state it was given with only the novelty set replaced."""
    engine, env, net, oenv, onet = _c4_setup(oracle, eaz, 2, seed=4)
    B, S, T = 24, 12, 4
    search = engine.Search(env, net, B, S, discount=0.997, two_players_game=True)
    st = env.init(B, key=2)
    search.selfplay(st, T, key=31)                       # captures the step graph on the old weights
    params, state = eaz.params.init_resnet_params(6, 7, 2, 7, 64, 2, seed=123)
    bs = eaz.params.init_hash_state(fill=0.2, seed=5)[eaz.params.HASH_MODULE]["binary_set"]
    state[eaz.params.RESNET_HASH_MODULE] = {"binary_set": bs}
    assert net.sync_model(params, state) is True and net.sync_model(params, state) is False
    onet2 = oracle.OracleNet.resnet(6, 7, 2, 7, eaz.params.resnet_array_list(params, state, 2), 64, 2, bitset=bs)
    st = env.init(B, key=2)
    ost = oenv.init(B, key=(2, 0))
    out = search.selfplay(st, T, key=31)
    ref = oracle.selfplay(oenv, onet2, ost, T=T, S=S, key=(31, 0), two_player=True)
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"]) and np.array_equal(out["state"].cpu().numpy(), ref["state"])
    np.testing.assert_allclose(out["root_value"].cpu().numpy(), ref["root_value"], rtol=0, atol=ATOL)
    # model_state round trip: BatchNorm statistics untouched, the bit set is the device's (updated by hash_update)
    obs = oenv.observe(ost)
    net.hash_update(torch.from_numpy(obs).cuda())
    got = net.get_state()
    assert set(got) == set(state)
    for mod in state:
        if mod == eaz.params.RESNET_HASH_MODULE:
            continue
        for k in state[mod]:
            assert np.array_equal(got[mod][k], state[mod][k]), (mod, k)
    new_bs = got[eaz.params.RESNET_HASH_MODULE]["binary_set"]
    assert (new_bs | bs == new_bs).all() and (new_bs != bs).any()
    assert net.sync_model(params, got) is False          # feeding the returned state back is a no-op


def test_hash_update_keeps_every_bit_of_a_shared_byte(oracle, eaz, torch):
    """Documented deviation: BaseHash.update is `binary_set.at[byte].set(binary_set[byte] | bit)` (hashes.py:45-50);
    when two samples of one batch fall into the same byte XLA's scatter keeps ONE of the two writes (unspecified which), so one bit can
    be lost.  The engine ORs atomically and keeps both -- a superset of whatever the reference stores.  A tiny set (2^8 bits) forces
    the collision."""
    from e_alphazero_amd import engine

    G, B = 8, 64
    am, params, _ = deepsea_setup(eaz, G, seed=1, fill=0.0)
    env = engine.EnvHandle.deepsea(G, am, device=0)
    net = engine.FCNet(env.ctx, G * G, 256, 2, params, None, hash_bits=8)
    obs = (np.random.default_rng(1).random((B, G * G)) < 0.2).astype(np.uint8)
    _, idx = net.hash_lookup(torch.from_numpy(obs).cuda())
    idx = idx.cpu().numpy().astype(np.int64)
    byte, bit = idx >> 3, idx & 7
    shared = [b for b in np.unique(byte) if len(np.unique(bit[byte == b])) > 1]
    assert shared, "no two samples share a byte: enlarge the batch"
    net.hash_update(torch.from_numpy(obs).cuda())
    bs = net.get_state()[eaz.params.HASH_MODULE]["binary_set"]
    expect = np.zeros(32, np.uint8)
    np.bitwise_or.at(expect, byte, (1 << bit).astype(np.uint8))
    assert np.array_equal(bs, expect)                    # every bit of every sample, shared bytes included
    seen, _ = net.hash_lookup(torch.from_numpy(obs).cuda())
    assert seen.all()


def test_selfplay_rejects_mismatched_output_buffers(eaz, torch):
    from e_alphazero_amd import engine

    G, B, S, T = 8, 16, 8, 3
    am, params, state = deepsea_setup(eaz, G, seed=1, fill=0.0)
    env = engine.EnvHandle.deepsea(G, am, device=0)
    search = engine.Search(env, engine.FCNet(env.ctx, G * G, 256, 2, params, state), B, S)
    st = env.init(B, key=0)
    out = search.selfplay(st, T, key=1)
    search.selfplay(st, T, key=2, outputs=out)           # same extents: reused
    with pytest.raises(eaz.EazError, match="outputs"):
        search.selfplay(st, T + 1, key=3, outputs=out)   # would be an out-of-bounds device write
    bad = dict(out, root_value=out["root_value"].double())
    with pytest.raises(eaz.EazError, match="root_value"):
        search.selfplay(st, T, key=3, outputs=bad)


# ---- SimHash (config.hash_class = "SimHash", src/network/hashes.py:100-139) ---------------------------------------------------------
@pytest.mark.parametrize("kind", ["deepsea", "subleq"])
def test_simhash_lookup_update_and_selfplay_match_oracle(oracle, eaz, torch, kind):
    """The second hash class end to end: indices, lookup / update, the forward's novelty, and self-play (incl. the fused DeepSea paths
    whose one-hot index table is rebuilt for SimHash) against the oracle with the same projection matrix."""
    from e_alphazero_amd import engine

    if kind == "deepsea":
        G, A = 8, 2
        am, params, _ = deepsea_setup(eaz, G, seed=2, fill=0.0)
        env = engine.EnvHandle.deepsea(G, am, device=0)
        oenv = oracle.OracleEnv.deepsea(G, am)
        in_dim, hash_off = G * G, 0
    else:
        W, A = 16, 16
        dims, params, _ = subleq_setup(eaz, W, seed=2, fill=0.0)
        env = engine.EnvHandle.subleq(W, (1,), True, device=0)
        oenv = oracle.OracleEnv.subleq(W, (1,), True)
        in_dim, hash_off = dims["in_dim"], dims["hash_off"]
    hash_len = in_dim - hash_off
    st = eaz.params.init_simhash_state(hash_len, fill=0.3, seed=5)[eaz.params.SIMHASH_MODULE]
    net = engine.FCNet(env.ctx, in_dim, 256, A, params, None, hash_off=hash_off, hash_len=hash_len)
    net.set_simhash(st["random_matrix"])
    net.set_state({eaz.params.SIMHASH_MODULE: st})
    onet = oracle.OracleNet(in_dim, 256, A, eaz.params.fc_weight_list(params), hash_off=hash_off, hash_len=hash_len, bitset=st["binary_set"].copy())
    onet.set_simhash(st["random_matrix"])
    B = 300
    ost, _ = oracle.random_play(oenv, B, 9, key=(4, 0))
    obs = oenv.observe(ost)
    seen, idx = net.hash_lookup(torch.from_numpy(obs).cuda())
    oidx = np.empty(B, np.uint32)
    hx = np.ascontiguousarray(obs[:, hash_off:])
    oracle.lib().eazo_simhash_indices(hx.ctypes.data, B, hash_len, 24, st["random_matrix"].ctypes.data, oidx.ctypes.data)
    assert np.array_equal(idx.cpu().numpy().view(np.uint32), oidx)
    out, ref = net.forward(torch.from_numpy(obs).cuda()), onet.forward(obs)
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=0, atol=ATOL)
    # self-play on the engine vs the oracle with the same hash class
    Bs, S, T = 40, 16, 6
    search = engine.Search(env, net, Bs, S, discount=0.97)
    dst = env.init(Bs, key=1)
    ost2 = oenv.init(Bs, key=(1, 0)) if kind == "subleq" else oenv.init(Bs)
    o = search.selfplay(dst, T, key=17)
    r = oracle.selfplay(oenv, onet, ost2, T=T, S=S, key=(17, 0), discount=0.97)
    assert np.array_equal(o["action"].cpu().numpy(), r["action"])
    for k in ("root_value", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(o[k].cpu().numpy(), r[k], rtol=0, atol=ATOL, err_msg=k)
    # update: scatter-OR of the SimHash indices; the state round trip carries the matrix
    net.hash_update(torch.from_numpy(obs).cuda())
    got = net.get_state()[eaz.params.SIMHASH_MODULE]
    expect = st["binary_set"].copy()
    np.bitwise_or.at(expect, (oidx >> 3).astype(np.int64), (1 << (oidx & 7)).astype(np.uint8))
    assert np.array_equal(got["binary_set"], expect) and np.array_equal(got["random_matrix"], st["random_matrix"])


@pytest.mark.parametrize("kind,H", [("deepsea", 128), ("deepsea", 30), ("subleq", 64), ("subleq", 512)])
def test_other_linear_layer_sizes(oracle, eaz, torch, kind, H):
    """config.linear_layer_size other than 256 (config.py:24): the forward of all four heads and a self-play call against the oracle.  H = 30:
    not a multiple of 4, i.e. the third layer is ONE chain (oracle o_linear_parts).  These sizes run fc_generic_forward_kernel on the
    per-simulation path (the one-launch kernels are specialised for 256)."""
    from e_alphazero_amd import engine

    if kind == "deepsea":
        G, A = 8, 2
        am, _, state = deepsea_setup(eaz, G, seed=5, fill=0.4)
        in_dim = G * G
        params = eaz.params.init_fc_params(in_dim, H, A, seed=H)
        env = engine.EnvHandle.deepsea(G, am, device=0)
        net = engine.FCNet(env.ctx, in_dim, H, A, params, state)
        oenv = oracle.OracleEnv.deepsea(G, am)
        onet = oracle.OracleNet(in_dim, H, A, eaz.params.fc_weight_list(params), bitset=state[eaz.params.HASH_MODULE]["binary_set"])
        disc, kw = 0.997, {}
    else:
        W = 16
        dims, _, state = subleq_setup(eaz, W, seed=3, fill=0.4)
        in_dim, A = dims["in_dim"], W
        params = eaz.params.init_fc_params(in_dim, H, A, seed=H)
        env = engine.EnvHandle.subleq(W, (1,), True, device=0)
        net = engine.FCNet(env.ctx, in_dim, H, A, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
        oenv = oracle.OracleEnv.subleq(W, (1,), True)
        onet = oracle.OracleNet(in_dim, H, A, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"], hash_len=dims["hash_len"],
                                bitset=state[eaz.params.HASH_MODULE]["binary_set"])
        disc, kw = 0.97, {}
    B, S, T = 37, 12, 4
    st = env.init(B, key=2)
    ost = oenv.init(B, key=(2, 0)) if kind == "subleq" else oenv.init(B)
    obs = oenv.observe(ost)
    obs[::3] = (np.random.default_rng(H).random(obs[::3].shape) < 0.3).astype(np.uint8)  # dense rows as well
    for a_, b_, name in zip(net.forward(torch.from_numpy(obs).cuda()), onet.forward(obs), ("exploit", "explore", "value", "ube", "novelty")):
        np.testing.assert_allclose(a_.cpu().numpy(), b_, rtol=0, atol=ATOL, err_msg=name)
    search = engine.Search(env, net, B, S, discount=disc)
    assert not search.launch_info()["one_launch"]
    out = search.selfplay(st, T, key=91, **kw)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(91, 0), discount=disc)
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    conv = (lambda a: subleq_device_to_oracle(a, 16)) if kind == "subleq" else (lambda a: a)
    assert np.array_equal(conv(out["state"].cpu().numpy()), ref["state"])
    for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)


def test_unsupported_shapes_fail_loudly(eaz, torch, kernel_variant):
    """linear_layer_size is 1..1024 (256, config.py:24, runs the MFMA kernels; any other size the plain kernel); Subleq word sizes are
    16..256 (subleq.py:606): anything else must be refused with a message, never computed on some fallback."""
    from e_alphazero_amd import engine

    env = engine.EnvHandle.deepsea(8, np.zeros((8, 8), np.uint8), device=0)
    with pytest.raises(eaz.EazError, match="linear_layer_size"):
        engine.FCNet(env.ctx, 64, 2048, 2, eaz.params.init_fc_params(64, 2048, 2, seed=0))
    with pytest.raises(eaz.EazError, match=r"\[16, 256\]"):
        engine.EnvHandle.subleq(8, (1,), True, device=0)
    with pytest.raises(eaz.EazError, match="multiple of 4"):
        engine.FCNet(env.ctx, 64, 256, 2, eaz.params.init_fc_params(64, 256, 2, seed=0), hash_off=2, hash_len=61)
    net = engine.FCNet(env.ctx, 64, 256, 2, eaz.params.init_fc_params(64, 256, 2, seed=0))
    with pytest.raises(eaz.EazError, match="observation size"):
        engine.Search(engine.EnvHandle.deepsea(4, np.zeros((4, 4), np.uint8), device=0), net, 8, 4)
    # a layer-1 input order is meaningless for one-hot observations: the DeepSea paths that feed the network an index instead of a row (the
    # one-launch kernel, the fused leaf step) refuse a network that carries one; the unfused variants run dense rows and simply follow it
    net.set_input_order([(32, 32), (0, 32)])
    search = engine.Search(env, net, 8, 4)
    if kernel_variant in ("persist", "persist-step", "auto-fused"):
        with pytest.raises(eaz.EazError, match="one-hot"):
            search.selfplay(env.init(8), 1, key=0)
    else:
        search.selfplay(env.init(8), 1, key=0)
    net.set_input_order(None)
    search.selfplay(env.init(8), 1, key=0)


@pytest.mark.parametrize("W", [16, 32])
def test_subleq_lowest_bytes_reward(oracle, eaz, torch, W):
    """Subleq(reward_fn=lowest_bytes) (subleq.py:540-542) on both step kernels (one wavefront per environment at word_size 16, one lane
    otherwise): the hand-worked answers of tests/test_oracle_envs.py (1/4 for `subleq OUT IN 0`, 1/7 for the five-word program), and
    bit-identical states against the oracle on programs built around the solving instruction at every cursor position."""
    from e_alphazero_amd import api, engine
    from helpers import ORACLE_SUBLEQ

    IN, OUT = W - 3, W - 2
    for prog, want in (([OUT, IN], np.float32(1.0) / np.float32(4.0)), ([0, 0, 3, OUT, IN], np.float32(1.0) / np.float32(7.0))):
        for fn, solved_reward in (("lowest_bytes", want), ("solved_or_not", np.float32(1.0))):
            env = engine.EnvHandle.subleq(W, (1,), True, device=0, reward_fn=fn)
            st = env.init(4, key=1)
            for t, word in enumerate(prog):
                st = env.step(st, torch.full((4,), word, dtype=torch.int32, device="cuda"), key=t)
                r = env.fields(st)["rewards"].cpu().numpy().reshape(-1)
                assert (r == (solved_reward if t == len(prog) - 1 else 0.0)).all(), (W, prog, fn, t, r)
    # programs `subleq 0 0 c` ... `subleq OUT IN c` at cursor c = 3 k: bytes_used = c + 3, every reachable denominator
    env = engine.EnvHandle.subleq(W, (1,), True, device=0, reward_fn="lowest_bytes")
    oenv = oracle.OracleEnv.subleq(W, (1,), True, reward_fn="lowest_bytes")
    cs = [c for c in range(3, W - 6, 3)]
    B = len(cs)
    ost = oenv.init(B, key=(5, 0))
    o = ost.view(ORACLE_SUBLEQ).reshape(B)
    mem = np.zeros((B, 256), np.int32)
    for i, c in enumerate(cs):
        mem[i, 2] = c
        mem[i, c], mem[i, c + 1] = OUT, IN       # word c + 2 = c arrives as the action below
    o["memory"] = mem
    o["step_count"] = np.array(cs) + 2
    ost = o.view(np.uint8).reshape(B, -1)
    st = torch.from_numpy(subleq_oracle_to_device(ost)).cuda()
    a = np.array(cs, np.int32)
    st = env.step(st, torch.from_numpy(a).cuda(), key=7)
    ost = oenv.step(ost, a, key=(7, 0))
    assert np.array_equal(subleq_device_to_oracle(st.cpu().numpy(), W), ost)
    oo = ost.view(ORACLE_SUBLEQ).reshape(B)
    solved = oo["solved"].astype(bool)
    assert solved.all()
    want = np.where(solved, np.float32(1.0) / (1 + (np.array(cs) + 3)).astype(np.float32), np.float32(0.0))
    assert np.array_equal(oo["reward"], want.astype(np.float32))
    # the facade takes the reference's functions (or their names) and refuses anything else
    assert api.Subleq([1], 16, api.lowest_bytes, True).reward_fn == "lowest_bytes"
    assert api.Subleq([1], 16, use_binary_encoding=True).reward_fn == "solved_or_not"
    with pytest.raises(eaz.EazError):
        api.Subleq([1], 16, reward_fn=lambda r: 1.0)



