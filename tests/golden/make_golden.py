#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/eaz_oracle.c) with fixed seeds.

The reference cannot run here (jax/pgx/emctx/haiku absent and unpinned), so these vectors pin
GPU-vs-oracle and oracle-vs-itself (regression) parity; they do NOT pin oracle-vs-JAX parity (**parity unpinned**).
Every case stores only seeds/inputs that cannot be regenerated plus the expected outputs; weights, action maps and
bit sets are regenerated from their seeds by tests/helpers.py.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import e_alphazero_amd as eaz  # noqa: E402
import eaz_oracle as O  # noqa: E402
from helpers import deepsea_setup, oracle_deepsea, subleq_setup  # noqa: E402

TREE_INT = ["node_visits", "parents", "action_from_parent", "children_index", "children_visits"]
TREE_F32 = ["node_values", "node_values_epistemic_variance", "children_values", "children_values_epistemic_variance",
            "children_prior_logits", "children_rewards", "children_discounts", "raw_values"]

CASES = {
    # name: (env kind, size, B, S, T, seed, fill, flags)
    "c1_deepsea8": ("deepsea", 8, 64, 16, 8, 0, 0.0, {}),                      # BASELINE.json configs[0] shapes
    "c2r_deepsea32": ("deepsea", 32, 32, 64, 3, 1, 0.5, {}),                   # configs[1] reduced in batch/steps
    "c1b_deepsea8_beta": ("deepsea", 8, 48, 16, 6, 2, 0.5, dict(directed_exploration=True, exploration_beta=1.0)),
    # configs[2] reduced; T = 16 real steps: Subleq-16 episodes end at _step_count >= 13 (subleq.py:671-673), so every environment is searched
    # from a terminated root and auto-reset (selfplay.py:26-75,135) inside the window
    "c3r_subleq16": ("subleq", 16, 24, 32, 16, 3, 0.25, {}),
    "c3b_subleq16_beta": ("subleq", 16, 24, 32, 16, 4, 0.25, dict(directed_exploration=True, exploration_beta=1.0)),
    # the same under the numerics switch "layer-1 constants first" (include/eaz.h eaz_net_set_input_order, oracle eazo_net_set_input_order):
    # every kernel variant has to follow the order, the one-launch kernel starts its layer-1 chains from per-tree partial sums
    "c3k_subleq16_constants_first": ("subleq", 16, 24, 32, 16, 6, 0.25, dict(directed_exploration=True, exploration_beta=1.0, constants_first=True)),
    # configs[3] reduced: connect_four, two-player discount, residual network with `size` blocks; T = 26: 10 of the 12 games end (win or
    # full column -> illegal move) and restart inside the window
    "c4r_connect4": ("connect4", 1, 12, 16, 26, 5, 0.3, dict(two_player=True)),
}


def build(kind, size, seed, fill, constants_first=False):
    if kind == "deepsea":
        am, params, state = deepsea_setup(eaz, size, seed=seed, fill=fill)
        env, net = oracle_deepsea(O, eaz, size, am, params, state)
        return env, net, 0.997
    if kind == "connect4":
        params, state = eaz.params.init_resnet_params(6, 7, 2, 7, 64, size, seed=seed)
        bs = eaz.params.init_hash_state(fill=fill, seed=seed)[eaz.params.HASH_MODULE]["binary_set"]
        env = O.OracleEnv.connect4()
        net = O.OracleNet.resnet(6, 7, 2, 7, eaz.params.resnet_array_list(params, state, size), 64, size, bitset=bs)
        return env, net, 0.997
    dims, params, state = subleq_setup(eaz, size, seed=seed, fill=fill)
    env = O.OracleEnv.subleq(size, (1,), True)
    net = O.OracleNet(dims["in_dim"], 256, size, eaz.params.fc_weight_list(params), hash_off=dims["hash_off"],
                      hash_len=dims["hash_len"], bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    if constants_first:
        net.set_input_order(eaz.params.subleq_constants_first(size, True))
    return env, net, 0.97  # config.py:122 discount for subleq


def main():
    for name, (kind, size, B, S, T, seed, fill, flags) in CASES.items():
        flags = dict(flags)
        env, net, discount = build(kind, size, seed, fill, flags.pop("constants_first", False))
        st = env.init(B, key=(seed, 0))
        out = O.selfplay(env, net, st, T=T, S=S, key=(1000 + seed, 0), discount=discount, **flags)
        # one extra search from the final states with a full tree dump
        A = env.num_actions
        ex, xp, v, u, _ = net.forward(env.observe(st))
        explore = bool(flags.get("directed_exploration", False))
        beta = (flags.get("exploration_beta", 0.0) * np.linspace(0, 1, B)).astype(np.float32)
        g = O.gumbel((2000 + seed, 0), B * A).reshape(B, A)
        act, aw, summ, tree = O.search(env, net, st, S, xp if explore else ex, v, u, beta, g, discount=discount,
                                       use_exploration_logits=explore, two_player=bool(flags.get("two_player", False)))
        d = tree.dump()
        data = dict(meta=np.array([size, B, S, T, seed], np.int64), fill=np.float64(fill), discount=np.float32(discount),
                    last_states=st, root_logits=xp if explore else ex, root_value=v, root_var=u, beta=beta, gumbel=g,
                    search_action=act, search_action_weights=aw)
        data.update({f"sp_{k}": v_ for k, v_ in out.items()})
        data.update({f"sum_{k}": v_ for k, v_ in summ.items()})
        data.update({f"tree_{k}": d[k] for k in TREE_INT + TREE_F32})
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **data)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
