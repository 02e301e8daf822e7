#!/usr/bin/env python3
"""Phase breakdown of fc16_forward_kernel<2> (one-hot forward inside a fused DeepSea search) from the diagnostic build
(make -C e-alphazero_amd/csrc diag).  Reads the s_memtime stamps of the LAST launch: per-phase medians over the workgroups and
the spread of start/end times across the grid."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EAZ_LIB", os.path.join(ROOT, "e-alphazero_amd", "libeaz_diag.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import e_alphazero_amd as eaz  # noqa: E402
from e_alphazero_amd import engine  # noqa: E402

G, B, S = 32, 4096, 64
am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
params = eaz.params.init_fc_params(G * G, 256, 2, seed=0)
env = engine.EnvHandle.deepsea(G, am, device=0)
net = engine.FCNet(env.ctx, G * G, 256, 2, params, eaz.params.init_hash_state(fill=0.5))
search = engine.Search(env, net, B, S, discount=0.997)
states = env.init(B, key=0)
search.selfplay(states, 3, key=0, use_graph=False)
# stop right after a simulation's forward: the step-wise API leaves the last fc16<2> launch in the stamp buffer
obs = env.observe(states)
ex, xp, v, u, _ = net.forward(obs)
search.begin(ex, v, u, states, torch.zeros(B, device="cuda"), None, torch.zeros(B, 2, device="cuda"))
for sim in range(40):
    search.simulate(sim)
torch.cuda.synchronize()
lib = eaz.load_library()
lib.eaz_diag_fc_stamps.argtypes = [C.c_void_p]
buf = np.zeros(1024 * 8, np.uint64)
lib.eaz_diag_fc_stamps(buf.ctypes.data)
st = buf.reshape(1024, 8)[:768].astype(np.int64)
st = st[st[:, 3] > st[:, 0]]
t0 = st[:, 0].min()
names = ["gather W1 rows + H1 + barrier", "layer 2 (dense MFMA)", "store H2 + barrier", "layer 3 + epilogue (wave 0)"]
print(f"{len(st)} workgroups; start spread {np.ptp(st[:, 0])} cycles; last stamp at {st[:, 4].max() - t0} cycles after the first start")
d = np.diff(st[:, :5], axis=1)
for i, n in enumerate(names):
    print(f"   {n:32s} median {np.median(d[:, i]):8.0f}   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f} cycles")
print(f"   start offset of workgroups (cycles after first): median {np.median(st[:, 0] - t0):.0f}, p90 {np.percentile(st[:, 0] - t0, 90):.0f}")

lib.eaz_diag_tree_stamps.argtypes = [C.c_void_p]
buf = np.zeros(1024 * 8, np.uint64)
lib.eaz_diag_tree_stamps(buf.ctypes.data)
ts = buf.reshape(1024, 8)[:256].astype(np.int64)
ts = ts[ts[:, 4] > ts[:, 0]]
d = np.diff(ts[:, :5], axis=1)
print(f"tree_step_kernel<2,1>, simulation 39 (thread 0 of each of {len(ts)} workgroups):")
for i, n in enumerate(["expand (loads, recurrent fn, stores)", "backward (path fold) + fence", "select (all levels)", "fused env step + stores"]):
    print(f"   {n:38s} median {np.median(d[:, i]):8.0f}   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f} cycles")
print(f"   select depth of thread 0's tree: median {np.median(ts[:, 5]):.0f}, p90 {np.percentile(ts[:, 5], 90):.0f}, max {ts[:, 5].max()}")
print(f"   select cycles per level: {np.median(d[:, 2] / np.maximum(ts[:, 5], 1)):.0f}")
