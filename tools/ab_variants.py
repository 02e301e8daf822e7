#!/usr/bin/env python3
"""A/B of the kernel-selection knobs at C2 (DeepSea-32, B=4096, 64 sims): graph-replay throughput and per-class kernel time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import e_alphazero_amd as eaz
from e_alphazero_amd import engine

G, B, S, T = 32, int(os.environ.get("B", 4096)), 64, 16
am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
params = eaz.params.init_fc_params(G * G, 256, 2, seed=0)
hstate = eaz.params.init_hash_state(fill=0.5, seed=0)
env = engine.EnvHandle.deepsea(G, am, device=0)
net = engine.FCNet(env.ctx, G * G, 256, 2, params, hstate)
ref = None
for name, (tile, fused) in {"auto-fused": (0, 1), "tile16-unfused": (16, 0), "tile64-unfused": (64, 0)}.items():
    engine.tune(fc_tile=tile, fused_leaf=fused, tree_block=64, tree_spread=0)
    search = engine.Search(env, net, B, S, discount=0.997)
    states = env.init(B, key=0)
    search.selfplay(states, 2, key=0)
    torch.cuda.synchronize()
    states = env.init(B, key=0)
    t0 = time.perf_counter(); out = search.selfplay(states, T, key=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sig = tuple(float(out[k].double().sum()) for k in ("root_value", "ube_prediction", "q_values_epistemic_variance")) + (int(out["action"].sum()),)
    if ref is None: ref = sig
    print(f"{name:16s}: {B*T/dt:9.0f} env-steps/s   identical outputs: {sig == ref}")
    search.profile(True); search.profile_read(reset=True)
    search.selfplay(states, 4, key=2, use_graph=False); torch.cuda.synchronize()
    for k, v in search.profile_read(reset=True).items():
        print(f"      {k:7s} {v['ms']:8.2f} ms / {v['launches']:5d} launches = {1e3 * v['ms'] / max(v['launches'], 1):7.2f} us")
    del search
