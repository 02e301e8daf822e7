#!/usr/bin/env python3
"""subleq_step_kernel alone (env.step on B states, no search): fresh states (every program enters a period-1 loop at once) vs
states after k uniformly random actions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import e_alphazero_amd as eaz
from e_alphazero_amd import engine

W, B = int(os.environ.get("W", 16)), int(os.environ.get("B", 4096))
env = engine.EnvHandle.subleq(W, (1,), True, device=0)
def timed(states, act, n=20):
    env.step(states, act)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): env.step(states, act)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator(device="cpu").manual_seed(0)
states = env.init(B, key=0)
zero = torch.zeros(B, dtype=torch.int32, device="cuda")
print(f"W={W} B={B}  fresh states, action 0: {timed(states, zero):8.1f} us per launch")
for k in range(1, 13):
    act = torch.randint(0, W, (B,), generator=g, dtype=torch.int32).cuda()
    states = env.step(states, act)
    act2 = torch.randint(0, W, (B,), generator=g, dtype=torch.int32).cuda()
    print(f"   after {k:2d} random actions: {timed(states, act2):8.1f} us per launch")
