#!/usr/bin/env python3
"""A/B of tree_step_kernel's launch shape at C2 (workgroup size x lanes-per-tree spread): graph-replay throughput."""
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
for block in (256, 128, 64):
    for spread in (1, 2, 4, 8):
        engine.tune(tree_block=block, tree_spread=spread)
        search = engine.Search(env, net, B, S, discount=0.997)
        states = env.init(B, key=0)
        search.selfplay(states, 2, key=0)
        torch.cuda.synchronize()
        states = env.init(B, key=0)
        t0 = time.perf_counter(); out = search.selfplay(states, T, key=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        sig = tuple(float(out[k].double().sum()) for k in ("root_value", "ube_prediction", "q_values_epistemic_variance")) + (int(out["action"].sum()),)
        if ref is None: ref = sig
        search.profile(True); search.profile_read(reset=True)
        search.selfplay(states, 2, key=2, use_graph=False); torch.cuda.synchronize()
        p = search.profile_read(reset=True)
        print(f"block {block:3d} spread {spread}: {B*T/dt:9.0f} env-steps/s  tree {1e3*p['tree']['ms']/p['tree']['launches']:6.2f} us  net {1e3*p['net']['ms']/p['net']['launches']:6.2f} us  identical: {sig == ref}", flush=True)
        del search
