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


# ---- Subleq state blobs: oracle layout (int32 memory[256], int32 io) <-> device layout (include/eaz.h eaz_subleq_state): oracle/eaz_oracle.py ----
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "oracle"))
from eaz_oracle import DEVICE_SUBLEQ, ORACLE_SUBLEQ, subleq_device_to_oracle, subleq_oracle_to_device  # noqa: E402,F401
