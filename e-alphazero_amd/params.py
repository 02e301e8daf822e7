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


SIMHASH_MODULE = "fc_az_net/sim_hash"
RESNET_SIMHASH_MODULE = "az_resnet/sim_hash"


def init_simhash_state(hash_len: int, bits_per_hash: int = 24, fill: float = 0.0, seed: int = 0, module: str = SIMHASH_MODULE):
    """haiku state of SimHash (src/network/hashes.py:40-43,126-129): the bit set and the RandomNormal (stddev 1) projection matrix
    [hash_len, bits_per_hash]."""
    bs = init_hash_state(bits_per_hash, fill, seed)[HASH_MODULE]["binary_set"]
    rm = np.random.default_rng(seed + 1).standard_normal((hash_len, bits_per_hash)).astype(np.float32)
    return {module: {"binary_set": bs, "random_matrix": rm}}


def subleq_constants_first(word_size: int, use_binary_encoding: bool = True):
    """Layer-1 input order "constants first" for a Subleq observation (engine.FCNet(input_order=...), OracleNet.set_input_order): the
    observation is memory | example_input | input_after | example_output | output_after (src/envs/subleq.py:697-705), each row-word
    `cols` inputs wide (:59-60); example_input / example_output are constants of an episode (:398-501), so they go first:
    [(start, len), ...] = example_input, example_output, memory, input_after, output_after."""
    cols = (word_size - 1).bit_length() + 1 if use_binary_encoding else word_size + 1
    mem, row = word_size * cols, 8 * cols
    return [(mem, row), (mem + 2 * row, row), (0, mem), (mem + row, row), (mem + 3 * row, row)]


def fc_weight_list(params):
    """[(w, b)] * 12 in haiku creation order."""
    return [(np.asarray(params[fc_layer_name(i)]["w"], np.float32), np.asarray(params[fc_layer_name(i)]["b"], np.float32))
            for i in range(NUM_FC_LAYERS)]


def save_npz(path: str, params, state) -> None:
    """Checkpoint interop (main.py:325-340): one flat .npz, keys "params/<haiku module>/<leaf>" and
    "state/<haiku module>/<leaf>" -- exactly what the reference-side export of INTEGRATION.md ("Checkpoints") writes from
    dic["model"] = (params, state).  Every leaf is kept (BatchNorm EMAs carry `hidden` / `counter` beside `average`)."""
    flat = {}
    for prefix, tree in (("params", params), ("state", state or {})):
        for mod, d in tree.items():
            for k, v in d.items():
                flat[f"{prefix}/{mod}/{k}"] = np.asarray(v)
    np.savez(path, **flat)


_STATE_LEAVES = ("binary_set", "random_matrix")


def load_npz(path: str):
    """-> (params, state) haiku-style dicts.  Loaded with allow_pickle=False: nothing in the file is executed.  Files without the
    "params/" / "state/" prefixes (LEGACY round-1/2 layout only; save_npz and the INTEGRATION.md recipe always write the prefixes) are split by
    a heuristic: hash-module leaves and "/~/" (EMA) modules are state.  That is right for the reference's networks, whose parameter-holding
    modules are all created in __call__; a haiku module created inside another module's __init__ would also carry "/~/" in its PARAMETER
    path, so a legacy file in which a "/~/" module holds a `w` or `b` leaf is ambiguous and refused."""
    z = np.load(path, allow_pickle=False)
    params, state = {}, {}
    for key in z.files:
        head, _, rest = key.partition("/")
        if head in ("params", "state") and "/" in rest:
            mod, name = rest.rsplit("/", 1)
            (params if head == "params" else state).setdefault(mod, {})[name] = z[key]
        else:
            mod, name = key.rsplit("/", 1)
            if "/~/" in mod and name in ("w", "b"):
                raise ValueError(f"{path}: legacy (prefix-less) checkpoint key {key!r} is a weight leaf under a '/~/' module -- cannot tell "
                                 "parameters from state; re-export with the 'params/' / 'state/' prefixes (INTEGRATION.md, Checkpoints)")
            is_state = name in _STATE_LEAVES or "/~/" in mod
            (state if is_state else params).setdefault(mod, {})[name] = z[key]
    return params, state


# ------------------------------------------------------------------------------------------------------------------
# EpistemicResidualAZNet (src/network/resnet.py): haiku module names in creation order [EXT naming]
# ------------------------------------------------------------------------------------------------------------------
RESNET_PREFIX = "az_resnet"
RESNET_HASH_MODULE = "az_resnet/xxhash32"


def _suffix(base: str, i: int) -> str:
    return base if i == 0 else f"{base}_{i}"


def resnet_module_order(num_blocks: int):
    """[(kind, module name)] in the order resnet.py:72-121 creates them: kind in {"conv", "bn", "linear"}."""
    P = RESNET_PREFIX
    order = [("conv", f"{P}/conv2_d")]
    for b in range(num_blocks):
        order += [("bn", f"{P}/block_{b}/batch_norm"), ("conv", f"{P}/block_{b}/conv2_d"),
                  ("bn", f"{P}/block_{b}/batch_norm_1"), ("conv", f"{P}/block_{b}/conv2_d_1")]
    order += [("bn", f"{P}/batch_norm")]
    order += [("conv", f"{P}/conv2_d_1"), ("bn", f"{P}/batch_norm_1"), ("linear", f"{P}/linear")]                       # policy
    order += [("conv", f"{P}/conv2_d_2"), ("bn", f"{P}/batch_norm_2"), ("linear", f"{P}/linear_1")]                     # exploration
    order += [("conv", f"{P}/conv2_d_3"), ("bn", f"{P}/batch_norm_3"), ("linear", f"{P}/linear_2"), ("linear", f"{P}/linear_3")]  # value
    order += [("conv", f"{P}/conv2_d_4"), ("bn", f"{P}/batch_norm_4"), ("linear", f"{P}/linear_4"), ("linear", f"{P}/linear_5")]  # ube
    return order


def init_resnet_params(H: int, W: int, cin: int, num_actions: int, num_channels: int = 64, num_blocks: int = 5, seed: int = 0):
    """Random-init weights of the reference architecture (hk.Conv2D / hk.Linear: trunc-normal, sigma = 1/sqrt(fan_in), zero
    bias) and SYNTHETIC BatchNorm statistics (scale ~ U(0.8,1.2), offset ~ N(0,0.1), mean ~ N(0,0.1), var ~ U(0.5,1.5)): a
    never-trained haiku state has var_ema = 0, which would scale activations by 316x per BatchNorm.  -> (params, state)."""
    rng = np.random.default_rng(seed)

    def tn(shape, fan_in):
        w = rng.standard_normal(shape)
        bad = np.abs(w) > 2.0
        while bad.any():
            w[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(w) > 2.0
        return (w / np.sqrt(fan_in)).astype(np.float32)

    C, P = num_channels, H * W
    conv_shapes = [(3, 3, cin, C)] + [(3, 3, C, C)] * (2 * num_blocks) + [(1, 1, C, 2), (1, 1, C, 2), (1, 1, C, 1), (1, 1, C, 1)]
    bn_ch = [C] * (2 * num_blocks) + [C, 2, 2, 1, 1]
    lin_shapes = [(P * 2, num_actions), (P * 2, num_actions), (P, C), (C, 1), (P, C), (C, 1)]
    ci, bi, li = iter(conv_shapes), iter(bn_ch), iter(lin_shapes)
    params, state = {}, {RESNET_HASH_MODULE: {"binary_set": np.zeros(1 << 21, np.uint8)}}
    for kind, name in resnet_module_order(num_blocks):
        if kind == "conv":
            sh = next(ci)
            params[name] = {"w": tn(sh, sh[0] * sh[1] * sh[2]), "b": np.zeros(sh[3], np.float32)}
        elif kind == "linear":
            sh = next(li)
            params[name] = {"w": tn(sh, sh[0]), "b": np.zeros(sh[1], np.float32)}
        else:
            c = next(bi)
            params[name] = {"scale": rng.uniform(0.8, 1.2, c).astype(np.float32), "offset": (0.1 * rng.standard_normal(c)).astype(np.float32)}
            state[f"{name}/~/mean_ema"] = {"average": (0.1 * rng.standard_normal(c)).astype(np.float32)}
            state[f"{name}/~/var_ema"] = {"average": rng.uniform(0.5, 1.5, c).astype(np.float32)}
    return params, state


def resnet_array_list(params, state, num_blocks: int):
    """Flat list of float32 arrays in the order eaz_net_create_resnet expects (42 + 12*num_blocks arrays)."""
    out = []
    for kind, name in resnet_module_order(num_blocks):
        if kind in ("conv", "linear"):
            out += [np.ascontiguousarray(params[name]["w"], np.float32).reshape(-1), np.ascontiguousarray(params[name]["b"], np.float32)]
        else:
            out += [np.ascontiguousarray(params[name]["scale"], np.float32), np.ascontiguousarray(params[name]["offset"], np.float32),
                    np.ascontiguousarray(state[f"{name}/~/mean_ema"]["average"], np.float32),
                    np.ascontiguousarray(state[f"{name}/~/var_ema"]["average"], np.float32)]
    return out
