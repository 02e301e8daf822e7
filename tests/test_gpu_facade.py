"""The host-side mirror of the reference interface (e-alphazero_amd/api.py): this file re-writes src/selfplay.py's step_fn with
the facade's names and checks it against the fused native selfplay() and against the generic-recurrent-fn path."""
import functools

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("kernel_variant")]


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available()
    return t


def _setup(api, env, B, S, T, **cfg):
    config = api.Config(selfplay_batch_size=B, selfplay_simulations_per_step=S, selfplay_steps=T, **cfg)
    forward = api.get_forward_fn(env, config)
    model = forward.init(api.random.PRNGKey(3))
    context = api.Context(env=env, devices=[0], forward=forward,
                          selfplay_recurrent_fn=api.get_epistemic_recurrent_fn(
                              env=env, forward=forward, batch_size=B, exploration=config.directed_exploration,
                              discount=config.discount, two_players_game=config.two_players_game))
    return config, forward, model, context


def _reference_style_step(api, torch, model, config, context, states, gumbel_t, key):
    """src/selfplay.py:86-143 re-typed against the facade (explicit root noise instead of key1)."""
    model_params, model_state = model
    key1, key2, key3, key4 = api.random.split(key, num=4)
    (exploitation_logits, exploration_logits, value, value_epistemic_variance, _), _ = context.forward.apply(
        model_params, model_state, states.observation, is_training=False)
    selfplay_beta = config.exploration_beta if config.directed_exploration else 0.0
    policy_logits = exploration_logits if config.directed_exploration else exploitation_logits
    policy_logits = torch.ones_like(policy_logits) if config.uniform_search_policy else policy_logits
    root = api.EpistemicRootFnOutput(
        prior_logits=policy_logits, value=value, value_epistemic_variance=value_epistemic_variance, embedding=states,
        beta=selfplay_beta * torch.linspace(0, 1, value.numel(), device=value.device).reshape(value.shape))
    policy_output = api.epistemic_gumbel_muzero_policy(
        params=model, rng_key=key1, root=root, recurrent_fn=context.selfplay_recurrent_fn,
        num_simulations=config.selfplay_simulations_per_step, invalid_actions=~states.legal_action_mask,
        qtransform=functools.partial(api.epistemic_qtransform_completed_by_mix_value,
                                     rescale_values=config.rescale_q_values_in_search),
        gumbel=gumbel_t)
    keys = api.random.split(key2, states.shape[0])
    search_summary = policy_output.search_tree.epistemic_summary()
    chosen_action = policy_output.action
    next_state = api.vmap(api.auto_reset(context.env.step, context.env.init))(states, chosen_action, keys)
    return next_state, dict(action=chosen_action, root_value=search_summary.value,
                            root_epistemic_std=search_summary.value_epistemic_std, value_prediction=value,
                            ube_prediction=value_epistemic_variance,
                            q_values_epistemic_variance=search_summary.qvalues_epistemic_variance,
                            visit_probs=search_summary.visit_probs, action_weights=policy_output.action_weights)


@pytest.mark.parametrize("directed", [False, True])
def test_reference_call_sites_match_fused_selfplay(eaz, torch, directed):
    from e_alphazero_amd import api

    G, B, S, T = 8, 64, 16, 9
    env = api.DeepSea(size_of_grid=G, action_map_key=api.random.PRNGKey(1))
    assert env.num_actions == 2 and env.num_players == 1 and env.id == "deep_sea"
    config, forward, model, context = _setup(api, env, B, S, T, directed_exploration=directed, exploration_beta=1.0 if directed else 0.0)
    keys = api.random.split(api.random.PRNGKey(5), B)
    states = api.vmap(env.init)(keys)
    assert states.observation.shape == (B, G, G) and states.observation.dtype == torch.bool
    assert states.legal_action_mask.all() and not states.terminated.any() and states.rewards.shape == (B, 1)
    g = torch.randn(T, B, 2, device="cuda")
    last, out = api.selfplay(model, config, context, states, api.random.PRNGKey(11), gumbel=g)
    assert out.state.data.shape[:2] == (T, B) and out.q_values_epistemic_variance.shape == (T, B, 2)
    s = states
    for t in range(T):
        s, o = _reference_style_step(api, torch, model, config, context, s, g[t], api.random.fold_in(0, t))
        assert torch.equal(s.data, out.state.data[t]), f"state t={t}"
        for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
            assert torch.equal(o[k], getattr(out, k)[t]), f"{k} t={t}"
    assert torch.equal(s.data, last.data)
    # episodes are 7 steps long at G=8: the trajectory contains terminal states followed by reset states
    assert out.state.terminated.any() and (out.state._step_count == 0).any()


def test_config_linear_layer_size_through_the_facade(eaz, torch):
    """config.linear_layer_size (config.py:24) other than 256: get_forward_fn builds the FC net at that size, selfplay() runs the per-simulation
    kernels, and the reference-style step loop over the facade's own pieces reproduces it."""
    from e_alphazero_amd import api

    G, B, S, T = 8, 24, 8, 4
    env = api.DeepSea(size_of_grid=G, action_map_key=api.random.PRNGKey(2))
    config, forward, model, context = _setup(api, env, B, S, T, linear_layer_size=64)
    assert tuple(np.asarray(model[0]["fc_az_net/linear"]["w"]).shape) == (G * G, 64)  # the value head's first layer: [in, linear_layer_size]
    states = api.vmap(env.init)(api.random.split(api.random.PRNGKey(6), B))
    g = torch.randn(T, B, 2, device="cuda")
    last, out = api.selfplay(model, config, context, states, api.random.PRNGKey(12), gumbel=g)
    s = states
    for t in range(T):
        s, o = _reference_style_step(api, torch, model, config, context, s, g[t], api.random.fold_in(0, t))
        assert torch.equal(s.data, out.state.data[t]), f"state t={t}"
        for k in ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"):
            assert torch.equal(o[k], getattr(out, k)[t]), f"{k} t={t}"
    assert torch.equal(s.data, last.data)


def test_generic_recurrent_fn_path_matches_fused(eaz, torch):
    from e_alphazero_amd import api

    G, B, S = 8, 32, 16
    env = api.DeepSea(size_of_grid=G, action_map_key=7)
    config, forward, model, context = _setup(api, env, B, S, 4)
    states = api.vmap(env.init)(api.random.split(0, B))
    a = torch.randint(0, 2, (B,), device="cuda", dtype=torch.int32)
    states = api.vmap(env.step)(states, a, api.random.split(1, B))
    (ex, xp, v, u, _), _ = forward.apply(model[0], model[1], states.observation)
    root = api.EpistemicRootFnOutput(ex, v, u, states, torch.zeros_like(v))
    g = torch.randn(B, 2, device="cuda")
    native = context.selfplay_recurrent_fn
    fused = api.epistemic_gumbel_muzero_policy(model, 0, root, native, S, ~states.legal_action_mask, gumbel=g)
    fused_sum = fused.search_tree.epistemic_summary()
    fused_ci = fused.search_tree.children_index.clone()
    python_fn = lambda params, key, action, emb: native(params, key, action, emb)  # noqa: E731 (forces the generic path)
    generic = api.epistemic_gumbel_muzero_policy(model, 0, root, python_fn, S, ~states.legal_action_mask, gumbel=g)
    gsum = generic.search_tree.epistemic_summary()
    assert torch.equal(generic.action, fused.action)
    assert torch.equal(generic.search_tree.children_index, fused_ci)
    assert torch.equal(gsum.visit_counts, fused_sum.visit_counts)
    np.testing.assert_allclose(gsum.value.cpu().numpy(), fused_sum.value.cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose(generic.action_weights.cpu().numpy(), fused.action_weights.cpu().numpy(), atol=1e-5)


def test_subleq_facade_and_uniformrandomplay(eaz, torch):
    from e_alphazero_amd import api

    env = api.Subleq(tasks=[api.SubleqTask.NEGATION_POSITIVE], word_size=16, use_binary_encoding=True)
    B, S, T = 32, 16, 3
    config, forward, model, context = _setup(api, env, B, S, T, env_id="subleq-16", discount=0.97)
    assert forward.hash_off == 16 * 5  # subleq_hash_only_io (config.py:30): only the IO rows are hashed
    last, traj = api.uniformrandomplay(config, context, api.random.PRNGKey(2))
    assert traj.data.shape[:2] == (T, B) and last.observation.shape == (B, 48, 5)
    last2, out = api.selfplay(model, config, context, last, api.random.PRNGKey(3))
    assert out.root_value.shape == (T, B) and torch.isfinite(out.root_value).all()
    assert (out.ube_prediction >= 0).all() and (out.ube_prediction <= 1).all()
    with pytest.raises(AssertionError):
        api.Subleq(tasks=[1], word_size=8)


def test_subleq_facade_layer1_constants_first_switch(oracle, eaz, torch):
    """Config.layer1_constants_first (NOT a field of the reference: the numerics switch of include/eaz.h eaz_net_set_input_order) through the
    reference-named surface: api.selfplay on Subleq-16 with the switch on equals the oracle under eazo_net_set_input_order, forward.apply
    follows the same order, and the switch off gives the oracle's default-order result."""
    from e_alphazero_amd import api, engine
    from helpers import subleq_device_to_oracle

    W, B, S, T = 16, 24, 16, 4
    env = api.Subleq(tasks=[api.SubleqTask.NEGATION_POSITIVE], word_size=W, use_binary_encoding=True)
    for switch in (True, False):
        config, forward, model, context = _setup(api, env, B, S, T, env_id="subleq-16", discount=0.97, layer1_constants_first=switch)
        states0 = api.vmap(env.init)(api.random.split(api.random.PRNGKey(5), B))
        ost = subleq_device_to_oracle(states0.data.cpu().numpy(), W)
        _, out = api.selfplay(model, config, context, states0, api.random.PRNGKey(11))
        oenv = oracle.OracleEnv.subleq(W, (1,), True)
        onet = oracle.OracleNet(forward.in_dim, 256, W, eaz.params.fc_weight_list(model[0]), hash_off=forward.hash_off,
                                hash_len=forward.in_dim - forward.hash_off, bitset=model[1][eaz.params.HASH_MODULE]["binary_set"])
        if switch:
            onet.set_input_order(eaz.params.subleq_constants_first(W, True))
        assert (forward.net(model).input_order is not None) == switch
        # forward.apply on the initial observations
        obs = oenv.observe(ost)
        (ex, xp, v, u, nov), _ = forward.apply(model[0], model[1], torch.from_numpy(obs).cuda().reshape(B, 48, 5))
        oex, oxp, ov, ou, onov = onet.forward(obs)
        assert np.array_equal(ex.cpu().numpy(), oex) and np.array_equal(v.cpu().numpy(), ov) and np.array_equal(u.cpu().numpy(), ou)
        ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=engine.key_pair(api.random.PRNGKey(11)), discount=0.97)
        assert np.array_equal(subleq_device_to_oracle(out.state.data.cpu().numpy(), W), ref["state"]), switch
        assert np.array_equal(out.root_value.cpu().numpy(), ref["root_value"]), switch


def test_reanalyze_and_evaluate_facade(eaz, torch):
    from e_alphazero_amd import api

    G, B, S, T = 8, 32, 16, 10
    env = api.DeepSea(size_of_grid=G, action_map_key=3)
    config = api.Config(selfplay_batch_size=B, selfplay_simulations_per_step=S, selfplay_steps=T, reanalyze_batch_size=B,
                        reanalyze_simulations_per_step=S, num_eval_episodes=B)
    forward = api.get_forward_fn(env, config)
    model = forward.init(api.random.PRNGKey(0))
    mk = lambda explore: api.get_epistemic_recurrent_fn(env, forward, B, explore, config.discount, False)  # noqa: E731
    context = api.Context(env, [0], forward, mk(False), mk(False), mk(False))
    last, traj = api.uniformrandomplay(config, context, api.random.PRNGKey(4))
    pair = api.ExperiencePair(first=traj[3], second=traj[4])
    out = api.reanalyze(model, config, context, pair, api.random.PRNGKey(5))
    assert out.observation.shape == (B, G, G) and out.value_target.shape == (B,)
    assert torch.allclose(out.exploitation_policy_target.sum(-1), torch.ones(B, device="cuda"), atol=1e-5)
    assert torch.allclose(out.exploration_policy_target.sum(-1), torch.ones(B, device="cuda"), atol=1e-5)
    assert (out.ube_target >= 0).all()
    ret = api.evaluate(model, config, context, api.random.PRNGKey(6))
    assert ret.ndim == 0 and 0.0 <= float(ret) <= 1.0


def test_connect_four_facade(eaz, torch):
    """configs[3] through the reference-named surface: pgx.make("connect_four"), ResNet forward, two-player search."""
    from e_alphazero_amd import api

    env = api.make("connect_four")
    assert env.num_players == 2 and env.num_actions == 7 and env.observation_shape == (6, 7, 2)
    B, S, T = 24, 16, 6
    config = api.Config(env_id="connect_four", two_players_game=True, selfplay_batch_size=B, selfplay_simulations_per_step=S, selfplay_steps=T)
    forward = api.get_forward_fn(env, config)
    assert forward.residual
    model = forward.init(api.random.PRNGKey(1))
    rf = api.get_epistemic_recurrent_fn(env, forward, B, False, config.discount, config.two_players_game)
    context = api.Context(env, [0], forward, rf)
    states = api.vmap(env.init)(api.random.split(2, B))
    assert states.observation.shape == (B, 6, 7, 2) and set(states.current_player.tolist()) <= {0, 1}
    (ex, xp, v, u, nov), _ = forward.apply(model[0], model[1], states.observation)
    assert ex.shape == (B, 7) and v.shape == (B,)
    # the native recurrent fn called with the emctx signature: reward of the player who moved, discount -gamma
    out, nxt = rf(model, 0, torch.full((B,), 3, dtype=torch.int32, device="cuda"), states)
    assert torch.allclose(out.discount, torch.full((B,), -config.discount, device="cuda")) and (out.reward == 0).all()
    assert (nxt.current_player == 1 - states.current_player).all()
    last, data = api.selfplay(model, config, context, states, api.random.PRNGKey(3))
    assert data.state.data.shape[:2] == (T, B) and (data.state._step_count[-1] == T).all()
    assert torch.isfinite(data.root_value).all()


def test_simhash_config_through_the_facade(eaz, torch):
    """config.hash_class = "SimHash" (config.py:27): the model_state carries '<net>/sim_hash' {binary_set, random_matrix}, the facade
    builds the device network with that projection matrix, and update_hash hands the state back under the same module name."""
    from e_alphazero_amd import api, params as P

    G, B, S, T = 8, 32, 8, 4
    env = api.DeepSea(size_of_grid=G, action_map_key=api.random.PRNGKey(1))
    config, forward, model, context = _setup(api, env, B, S, T, hash_class="SimHash")
    assert set(model[1]) == {P.SIMHASH_MODULE} and model[1][P.SIMHASH_MODULE]["random_matrix"].shape == (G * G, 24)
    states = api.vmap(env.init)(api.random.split(api.random.PRNGKey(5), B))
    last, out = api.selfplay(model, config, context, states, api.random.PRNGKey(11))
    assert out.state.data.shape[:2] == (T, B)
    assert (out.ube_prediction == 1.0).all()          # empty novelty set: every state is novel, u = max_u
    (_, _, _, u, nov), new_state = forward.apply(model[0], model[1], states.observation, update_hash=True)
    assert set(new_state) == {P.SIMHASH_MODULE} and new_state[P.SIMHASH_MODULE]["binary_set"].any()
    assert np.array_equal(new_state[P.SIMHASH_MODULE]["random_matrix"], model[1][P.SIMHASH_MODULE]["random_matrix"])
    (_, _, _, u2, nov2), _ = forward.apply(model[0], new_state, states.observation)
    assert (nov == 1.0).all() and (nov2 == 0.0).all()  # seen now
    with pytest.raises(eaz.EazError, match="LCGHash"):
        api.get_forward_fn(env, api.Config(hash_class="LCGHash"))
