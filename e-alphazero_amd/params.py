"""Model pytree helpers: the haiku layout the reference checkpoints use (`main.py:325-340`).

params: {"fc_az_net/linear": {"w": [in,out], "b": [out]}, "fc_az_net/linear_1": ..., "fc_az_net/linear_11": ...}
state : {"fc_az_net/xxhash32": {"binary_set": uint8[2**(bits-3)]}}

Module order = creation order in src/network/fully_connected.py:50-81:
value (0-2), ube (3-5), exploitation policy (6-8), exploration policy (9-11).
"""
from __future__ import annotations

import numpy as np

FC_PREFIX = "fc_az_net"
HASH_MODULE = "fc_az_net/xxhash32"
NUM_FC_LAYERS = 12


def fc_layer_name(i: int) -> str:
    return f"{FC_PREFIX}/linear" if i == 0 else f"{FC_PREFIX}/linear_{i}"


def fc_layer_dims(in_dim: int, hidden: int, num_actions: int):
    """(in, out) of the 12 hk.Linear modules."""
    dims = []
    for head in range(4):
        out = 1 if head < 2 else num_actions
        dims += [(in_dim, hidden), (hidden, hidden), (hidden, out)]
    return dims


def init_fc_params(in_dim: int, hidden: int, num_actions: int, seed: int = 0):
    """hk.Linear default init [EXT]: w ~ truncated-normal(+-2 sigma), sigma = 1/sqrt(fan_in); b = 0.

    Random-init weights of the reference architecture (there are no checkpoints to load offline).
    """
    rng = np.random.default_rng(seed)
    params = {}
    for i, (k, n) in enumerate(fc_layer_dims(in_dim, hidden, num_actions)):
        w = rng.standard_normal((k, n))
        bad = np.abs(w) > 2.0
        while bad.any():
            w[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(w) > 2.0
        params[fc_layer_name(i)] = {
            "w": (w / np.sqrt(k)).astype(np.float32),
            "b": np.zeros(n, np.float32),
        }
    return params


def init_hash_state(bits_per_hash: int = 24, fill: float = 0.0, seed: int = 0):
    """haiku state of XXHash (src/network/hashes.py:40-43): an all-zero (or randomly `fill`-ed) bit set."""
    n = 1 << (bits_per_hash - 3)
    if fill <= 0.0:
        bs = np.zeros(n, np.uint8)
    else:
        rng = np.random.default_rng(seed)
        bits = rng.random((n, 8)) < fill
        bs = np.packbits(bits, axis=1, bitorder="little").reshape(n)
    return {HASH_MODULE: {"binary_set": bs}}


def fc_weight_list(params):
    """[(w, b)] * 12 in haiku creation order."""
    return [(np.asarray(params[fc_layer_name(i)]["w"], np.float32), np.asarray(params[fc_layer_name(i)]["b"], np.float32))
            for i in range(NUM_FC_LAYERS)]


def save_npz(path: str, params, state) -> None:
    """Checkpoint interop: flat .npz with 'module/name' keys."""
    flat = {}
    for mod, d in list(params.items()) + list(state.items()):
        for k, v in d.items():
            flat[f"{mod}/{k}"] = np.asarray(v)
    np.savez(path, **flat)


def load_npz(path: str):
    z = np.load(path, allow_pickle=False)
    params, state = {}, {}
    for key in z.files:
        mod, name = key.rsplit("/", 1)
        (state if name == "binary_set" else params).setdefault(mod, {})[name] = z[key]
    return params, state
