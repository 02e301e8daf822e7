#!/usr/bin/env python3
"""Per-kernel timings on one MI355X (torch.cuda.Event on the current stream): FC forward on one-hot / dense observations,
novelty hash, a full search at C2 shapes with the library's per-class HIP-event profile."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import e_alphazero_amd as eaz  # noqa: E402
from e_alphazero_amd import engine  # noqa: E402


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def main():
    G, B, S = 32, 4096, 64
    am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
    params = eaz.params.init_fc_params(G * G, 256, 2, seed=0)
    st = eaz.params.init_hash_state(fill=0.5)
    env = engine.EnvHandle.deepsea(G, am, device=0)
    net = engine.FCNet(env.ctx, G * G, 256, 2, params, st)
    rng = np.random.default_rng(1)
    onehot = np.zeros((B, G * G), np.uint8)
    onehot[np.arange(B), rng.integers(0, G * G, B)] = 1
    same = np.zeros((B, G * G), np.uint8)
    same[:, 37] = 1
    dense = (rng.random((B, G * G)) < 0.3).astype(np.uint8)
    for name, obs in (("one-hot random", onehot), ("one-hot identical", same), ("dense 30%", dense)):
        o = torch.from_numpy(obs).cuda()
        t_all = timeit(lambda: net.forward(o))
        t_hash = timeit(lambda: net.hash_lookup(o))
        print(f"net.forward (hash + fc, 4 heads) {name:18s}: {t_all:8.1f} us   hash alone {t_hash:6.1f} us")
    search = engine.Search(env, net, B, S)
    states = env.init(B, key=0)
    search.selfplay(states, 2, key=0)
    torch.cuda.synchronize()
    search.profile(True)
    search.profile_read(reset=True)
    t0 = time.perf_counter()
    search.selfplay(states, 8, key=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = search.profile_read(reset=True)
    print(f"selfplay 8 steps: {dt * 1e3:.1f} ms -> {B * 8 / dt:.0f} env-steps/s")
    for k, v in p.items():
        print(f"  {k:7s} {v['ms']:8.2f} ms / {v['launches']:5d} launches = {1e3 * v['ms'] / max(v['launches'], 1):7.2f} us")


if __name__ == "__main__":
    main()
