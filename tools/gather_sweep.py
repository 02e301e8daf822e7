#!/usr/bin/env python3
"""The standalone leaf-gather kernel against the HBM roof as a function of the batch (north-star: ">= 40 % HBM roofline on the leaf-gather kernel").

In the production paths the gather does not exist as a launch (DeepSea / Subleq-16: the child state never leaves LDS; connect_four / wide Subleq: it is
part of the fused leaf step), and at BASELINE's batch sizes a launch moves 1-4 MB -- microseconds of HBM time, i.e. launch latency.  This tool runs the
UNFUSED DeepSea variant (EAZ_TUNE_FUSED_LEAF = 0: leaf_gather_deepsea_kernel gathers the parent state, steps it, writes the child state and emits the
1 KiB observation row as 16-B-per-lane stores) at growing batches and prints achieved bytes / kernel time / 8 TB/s: what the kernel reaches once a launch
carries enough bytes to be bandwidth-bound at all.  Event pairs on the launch stream (eaz_profile_*), S = 2 simulations, one real step.

usage (GPU box): python tools/gather_sweep.py [B ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import e_alphazero_amd as eaz  # noqa: E402
from e_alphazero_amd import engine  # noqa: E402

PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md
G, S = 32, 2
engine.tune(fused_leaf=0, persist=0, shards=1)
am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
params = eaz.params.init_fc_params(G * G, 256, 2, seed=0)
env = engine.EnvHandle.deepsea(G, am, device=0)
net = engine.FCNet(env.ctx, G * G, 256, 2, params, eaz.params.init_hash_state(fill=0.5))
rows = []
for B in [int(x) for x in sys.argv[1:]] or [4096, 16384, 65536, 262144, 1048576]:
    search = engine.Search(env, net, B, S)
    st = env.init(B, key=0)
    search.selfplay(st, 1, key=0, use_graph=False)          # warm-up
    search.profile(True)
    search.profile_read(reset=True)
    for i in range(3):
        search.selfplay(st, 1, key=1 + i, use_graph=False)
    torch.cuda.synchronize()
    prof = search.profile_read(reset=True)
    search.profile(False)
    n, ms = prof["gather"]["launches"], prof["gather"]["ms"]
    bytes_per_launch = B * (2 * env.state_bytes + env.obs_dim)   # compact layout: parent state in, child state + u8 observation row out
    us = 1e3 * ms / max(n, 1)
    gbps = bytes_per_launch / (us * 1e-6) / 1e9 if us > 0 else 0.0
    rows.append(dict(B=B, launches=n, us_per_launch=round(us, 2), MB_per_launch=round(bytes_per_launch / 1e6, 2), GBps=round(gbps, 1), frac_of_hbm_peak=round(gbps / PEAK, 3)))
    print(rows[-1], flush=True)
    del search
    torch.cuda.empty_cache()
print(json.dumps({"kernel": "leaf_gather_deepsea_kernel", "peak_GBps": PEAK, "bytes_per_tree_sim": 2 * env.state_bytes + env.obs_dim, "sweep": rows}))
