#!/usr/bin/env python3
"""Scratch driver: Subleq-16 search.run() / selfplay at a given batch (debugging aid for search_subleq.hpp)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import e_alphazero_amd as eaz  # noqa: E402
from e_alphazero_amd import engine  # noqa: E402

W, B, S = 16, int(os.environ.get("B", 4096)), 64
in_dim, hash_off = 240, 80
params = eaz.params.init_fc_params(in_dim, 256, W, seed=0)
env = engine.EnvHandle.subleq(W, (1,), True, device=0)
net = engine.FCNet(env.ctx, in_dim, 256, W, params, eaz.params.init_hash_state(fill=0.5), hash_off=hash_off, hash_len=in_dim - hash_off)
search = engine.Search(env, net, B, S, discount=0.97)
states = env.init(B, key=0)
print("init ok", flush=True)
for T in (1, 2, 6):
    t0 = time.perf_counter()
    search.selfplay(states, T, key=T, use_graph=False)
    torch.cuda.synchronize()
    print(f"selfplay T={T} ok {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for T in (2, 2):
    t0 = time.perf_counter()
    search.selfplay(states, T, key=T, use_graph=True)
    torch.cuda.synchronize()
    print(f"selfplay graph T={T} ok {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
