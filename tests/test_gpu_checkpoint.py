"""N4 checkpoint interop on the GPU (src/main.py:325-340): a model written as the flat .npz of INTEGRATION.md ("Checkpoints") -- the layout
the reference-side export produces from dic["model"] = (params, state) -- is loaded with numpy (allow_pickle=False), handed to the
facade's forward fn / the engine's networks, and drives selfplay(); the results must equal the oracle's on the same arrays.
FC layout with a non-trivial novelty set (binary_set), and the residual layout with BatchNorm EMA statistics and its own bit set.
"""
import numpy as np
import pytest

from helpers import subleq_device_to_oracle, subleq_setup

pytestmark = pytest.mark.gpu
ATOL = 1e-5
FLOATS = ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance")


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available()
    return t


def _compare(out, ref, state_cmp=lambda s: s):
    assert np.array_equal(out["action"].cpu().numpy(), ref["action"])
    assert np.array_equal(state_cmp(out["state"].cpu().numpy()), ref["state"])
    for k in FLOATS:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)


def test_fc_checkpoint_through_the_facade_and_the_engine(oracle, eaz, torch, tmp_path):
    """save_npz -> load_npz -> api.get_forward_fn(...).net(model) -> api.selfplay, and the same arrays through engine.FCNet directly."""
    from e_alphazero_amd import api, engine

    W, B, S, T = 16, 24, 16, 15
    dims, params, state = subleq_setup(eaz, W, seed=21, fill=0.35)
    path = str(tmp_path / "fc.npz")
    eaz.params.save_npz(path, params, state)
    with np.load(path, allow_pickle=False) as z:
        assert sorted(z.files)[0].startswith("params/fc_az_net/linear") and "state/fc_az_net/xxhash32/binary_set" in z.files
    p2, s2 = eaz.params.load_npz(path)
    assert all(np.array_equal(params[k][n], p2[k][n]) for k in params for n in params[k])
    assert np.array_equal(s2[eaz.params.HASH_MODULE]["binary_set"], state[eaz.params.HASH_MODULE]["binary_set"])

    oenv = oracle.OracleEnv.subleq(W, (1,), True)
    onet = oracle.OracleNet(dims["in_dim"], 256, W, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"], hash_len=dims["hash_len"],
                            bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    ost = oenv.init(B)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(91, 0), discount=0.97)

    # the facade (reference names): Subleq env, Config, forward fn, recurrent fn, selfplay
    env = api.Subleq(tasks=(api.SubleqTask.NEGATION_POSITIVE,), word_size=W, use_binary_encoding=True)
    config = api.Config(env_id="subleq", selfplay_batch_size=B, selfplay_simulations_per_step=S, selfplay_steps=T, discount=0.97)
    forward = api.get_forward_fn(env, config)
    context = api.Context(env=env, devices=[0], forward=forward,
                          selfplay_recurrent_fn=api.get_epistemic_recurrent_fn(env=env, forward=forward, batch_size=B, exploration=False,
                                                                               discount=0.97, two_players_game=False))
    keys = api.random.split(api.random.PRNGKey(0), B)
    last = api.vmap(env.init)(keys)
    last2, sp = api.selfplay((p2, s2), config, context, last, 91)
    out = dict(action=None, state=sp.state.data, root_value=sp.root_value, root_epistemic_std=sp.root_epistemic_std,
               value_prediction=sp.value_prediction, ube_prediction=sp.ube_prediction, q_values_epistemic_variance=sp.q_values_epistemic_variance)
    assert np.array_equal(subleq_device_to_oracle(out["state"].cpu().numpy(), W), ref["state"])
    for k in FLOATS:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=ATOL, err_msg=k)
    assert np.array_equal(subleq_device_to_oracle(last2.data.cpu().numpy(), W), ost)

    # the engine objects directly
    henv = engine.EnvHandle.subleq(W, (1,), True, device=0)
    net = engine.FCNet(henv.ctx, dims["in_dim"], 256, W, p2, s2, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
    search = engine.Search(henv, net, B, S, discount=0.97)
    st = henv.init(B, key=0)
    out = search.selfplay(st, T, key=91)
    _compare(out, ref, lambda s: subleq_device_to_oracle(s, W))
    # the novelty set travelled: the device's bit set is the checkpoint's
    assert np.array_equal(net.get_state()[eaz.params.HASH_MODULE]["binary_set"], state[eaz.params.HASH_MODULE]["binary_set"])


def test_resnet_checkpoint_with_batchnorm_state(oracle, eaz, torch, tmp_path):
    from e_alphazero_amd import engine

    B, S, T, nb = 10, 12, 9, 2
    params, state = eaz.params.init_resnet_params(6, 7, 2, 7, 64, nb, seed=33)
    bs = eaz.params.init_hash_state(fill=0.3, seed=33)[eaz.params.HASH_MODULE]["binary_set"]
    state[eaz.params.RESNET_HASH_MODULE] = {"binary_set": bs}
    # haiku's ExponentialMovingAverage keeps `hidden` and `counter` beside `average`: the export writes every leaf, the loader keeps them
    for k in list(state):
        if k.endswith("_ema"):
            state[k]["hidden"] = state[k]["average"] * 0.5
            state[k]["counter"] = np.array(7, np.int32)
    path = str(tmp_path / "resnet.npz")
    eaz.params.save_npz(path, params, state)
    p2, s2 = eaz.params.load_npz(path)
    assert set(p2) == set(params) and set(s2) == set(state)
    assert all(np.array_equal(state[k][n], s2[k][n]) for k in state for n in state[k])

    oenv = oracle.OracleEnv.connect4()
    onet = oracle.OracleNet.resnet(6, 7, 2, 7, eaz.params.resnet_array_list(params, state, nb), 64, nb, bitset=bs)
    env = engine.EnvHandle.connect4(device=0)
    net = engine.ResNet(env.ctx, 6, 7, 2, 7, p2, s2, num_blocks=nb)
    search = engine.Search(env, net, B, S, discount=0.997, two_players_game=True)
    st = env.init(B, key=4)
    ost = st.cpu().numpy().copy()
    out = search.selfplay(st, T, key=92)
    ref = oracle.selfplay(oenv, onet, ost, T=T, S=S, key=(92, 0), two_player=True)
    _compare(out, ref)
    gs = net.get_state()
    assert np.array_equal(gs[eaz.params.RESNET_HASH_MODULE]["binary_set"], bs)
    assert all(np.array_equal(gs[k]["average"], state[k]["average"]) for k in state if k.endswith("_ema"))
