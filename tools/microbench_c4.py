#!/usr/bin/env python3
"""C4 shapes (BASELINE.json configs[3]): connect_four, ResNet 64ch x 5 blocks, B=8192, 200 sims (S overridable)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import e_alphazero_amd as eaz
from e_alphazero_amd import engine

B, S, T, nb = int(os.environ.get("B", 8192)), int(os.environ.get("S", 200)), int(os.environ.get("T", 2)), 5
params, state = eaz.params.init_resnet_params(6, 7, 2, 7, 64, nb, seed=0)
env = engine.EnvHandle.connect4(device=0)
net = engine.ResNet(env.ctx, 6, 7, 2, 7, params, state, num_blocks=nb)
search = engine.Search(env, net, B, S, discount=0.997, two_players_game=True)
states = env.init(B, key=0)
obs = env.observe(states)
for _ in range(3): net.forward(obs)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): net.forward(obs)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
flops = B * 2 * (42 * 18 * 64 + 10 * 42 * 576 * 64)
print(f"resnet forward B={B}: {ms:.3f} ms -> {flops / ms / 1e9:.1f} TFLOP/s (tower convs only)")
search.selfplay(states, 1, key=0)
torch.cuda.synchronize()
t0 = time.perf_counter(); search.selfplay(states, T, key=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"connect_four B={B} S={S}: {T} real steps {dt*1e3:.1f} ms -> {B*T/dt:.0f} env-steps/s")
search.profile(True); search.profile_read(reset=True)
search.selfplay(states, 1, key=2, use_graph=False); torch.cuda.synchronize()
for k, v in search.profile_read(reset=True).items():
    print(f"  {k:7s} {v['ms']:8.2f} ms / {v['launches']:5d} launches = {1e3 * v['ms'] / max(v['launches'], 1):8.2f} us")
