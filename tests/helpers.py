"""Shared builders for the parity tests: same seeded inputs for the oracle and the HIP engine."""
import numpy as np


def deepsea_setup(eaz, G, seed=0, fill=0.0, A=2, H=256):
    am = (np.random.default_rng(seed).random((G, G)) < 0.5).astype(np.uint8)
    params = eaz.params.init_fc_params(G * G, H, A, seed=seed)
    state = eaz.params.init_hash_state(fill=fill, seed=seed)
    return am, params, state


def oracle_deepsea(O, eaz, G, am, params, state):
    env = O.OracleEnv.deepsea(G, am)
    net = O.OracleNet(G * G, 256, 2, eaz.params.fc_weight_list(params), bitset=state[eaz.params.HASH_MODULE]["binary_set"])
    return env, net


def subleq_setup(eaz, W, binary=True, hash_io=True, seed=0, fill=0.0):
    bl = (W - 1).bit_length()
    cols = bl + 1 if binary else W + 1
    rows = W + 32
    in_dim = rows * cols
    params = eaz.params.init_fc_params(in_dim, 256, W, seed=seed)
    state = eaz.params.init_hash_state(fill=fill, seed=seed)
    hash_off = W * cols if hash_io else 0
    return dict(in_dim=in_dim, rows=rows, cols=cols, hash_off=hash_off, hash_len=in_dim - hash_off), params, state


def random_states_deepsea(G, B, seed=0):
    """Random reachable compact DeepSea states as raw 16-byte blobs (step_count, pos <= step_count)."""
    rng = np.random.default_rng(seed)
    sc = rng.integers(0, G - 1, size=B).astype(np.int32)
    pos = (rng.random(B) * (sc + 1)).astype(np.int32)
    blob = np.zeros((B, 16), np.uint8)
    blob[:, 0:4] = sc.view(np.uint8).reshape(B, 4)
    blob[:, 4:8] = pos.view(np.uint8).reshape(B, 4)
    return blob


# ---- Subleq state blobs: oracle layout (int32 memory[256], int32 io) <-> device layout (include/eaz.h eaz_subleq_state) ----
ORACLE_SUBLEQ = np.dtype([("step_count", "<i4"), ("task", "<i4"), ("reward", "<f4"), ("terminated", "u1"), ("truncated", "u1"),
                          ("solved", "u1"), ("pad", "u1"), ("memory", "<i4", 256), ("ex_in", "<i4", 8), ("ex_in_after", "<i4", 8),
                          ("ex_out", "<i4", 8), ("ex_out_after", "<i4", 8)])
DEVICE_SUBLEQ = np.dtype([("step_count", "<i4"), ("task", "<i4"), ("reward", "<f4"), ("terminated", "u1"), ("truncated", "u1"),
                          ("solved", "u1"), ("pad", "u1"), ("ex_in", "<u2", 8), ("ex_in_after", "<u2", 8), ("ex_out", "<u2", 8),
                          ("ex_out_after", "<u2", 8), ("memory", "u1", 256)])
assert ORACLE_SUBLEQ.itemsize == 1168 and DEVICE_SUBLEQ.itemsize == 336


def subleq_oracle_to_device(blobs):
    o = np.ascontiguousarray(blobs).view(ORACLE_SUBLEQ).reshape(blobs.shape[:-1])
    d = np.zeros(o.shape, DEVICE_SUBLEQ)
    for k in DEVICE_SUBLEQ.names:
        d[k] = o[k]
    return d.view(np.uint8).reshape(*o.shape, 336)


def subleq_device_to_oracle(blobs, W):
    d = np.ascontiguousarray(blobs).view(DEVICE_SUBLEQ).reshape(blobs.shape[:-1])
    o = np.zeros(d.shape, ORACLE_SUBLEQ)
    for k in DEVICE_SUBLEQ.names:
        o[k] = d[k]
    o["memory"][..., W:] = 0
    return o.view(np.uint8).reshape(*d.shape, 1168)
