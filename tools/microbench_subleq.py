#!/usr/bin/env python3
"""C3 shapes: Subleq word_size W, B=4096, 64 sims: per-class kernel time (instrumented) and graph-replay throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import e_alphazero_amd as eaz
from e_alphazero_amd import engine
from helpers import subleq_setup

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, S, T = int(os.environ.get("B", 4096)), 64, 8
explore = os.environ.get("BETA", "0") != "0"
dims, params, state = subleq_setup(eaz, W, seed=0, fill=0.5)
env = engine.EnvHandle.subleq(W, (1,), True, device=0)
net = engine.FCNet(env.ctx, dims["in_dim"], 256, W, params, state, hash_off=dims["hash_off"], hash_len=dims["hash_len"])
search = engine.Search(env, net, B, S, discount=0.97, exploration=explore)
kw = dict(directed_exploration=explore, exploration_beta=1.0 if explore else 0.0)
states = env.init(B, key=0)
search.selfplay(states, 2, key=0, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter(); search.selfplay(states, T, key=1, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"subleq-{W} B={B} S={S} beta={'1' if explore else '0'}: graph replay {T} steps {dt*1e3:.1f} ms -> {B*T/dt:.0f} env-steps/s")
search.profile(True); search.profile_read(reset=True)
t0 = time.perf_counter(); search.selfplay(states, T, key=2, use_graph=False, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
p = search.profile_read(reset=True)
print(f"instrumented: {dt*1e3:.1f} ms")
for k, v in p.items():
    print(f"  {k:7s} {v['ms']:8.2f} ms / {v['launches']:5d} launches = {1e3 * v['ms'] / max(v['launches'], 1):7.2f} us")
