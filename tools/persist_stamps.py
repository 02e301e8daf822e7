#!/usr/bin/env python3
"""Phase breakdown of search_persist_kernel at C2 from the diagnostic build (make -C e-alphazero_amd/csrc diag): s_memtime stamps
of simulation S/2, thread 0 (tree lane / MFMA wave 0) and thread 256 (helper wave) of every workgroup."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EAZ_LIB", os.path.join(ROOT, "e-alphazero_amd", "libeaz_diag.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import e_alphazero_amd as eaz  # noqa: E402
from e_alphazero_amd import engine  # noqa: E402
from e_alphazero_amd import _lib  # noqa: E402

if _lib.build_hash() != _lib.source_hash():  # the stamps must describe the kernels of THIS tree (make -C e-alphazero_amd/csrc diag)
    sys.exit(f"the diagnostic build is stale: {os.environ['EAZ_LIB']} was built from sources {_lib.build_hash()}, the tree is {_lib.source_hash()}")
print(f"diagnostic build of sources {_lib.build_hash()}")

G, B, S = 32, int(os.environ.get("B", 4096)), 64
am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
params = eaz.params.init_fc_params(G * G, 256, 2, seed=0)
env = engine.EnvHandle.deepsea(G, am, device=0)
net = engine.FCNet(env.ctx, G * G, 256, 2, params, eaz.params.init_hash_state(fill=0.5))
search = engine.Search(env, net, B, S, discount=0.997)
states = env.init(B, key=0)
search.selfplay(states, 12, key=0, use_graph=False)   # mid-episode states
obs = env.observe(states)
ex, xp, v, u, _ = net.forward(obs)
g = torch.zeros(B, 2, device="cuda")
for rep in range(3):
    search.begin(ex, v, u, states, torch.zeros(B, device="cuda"), None, g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    search.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"search.run(): {1e3 * dt:.3f} ms = {1e6 * dt / S:.2f} us per simulation (host-timed, one launch)")
lib = eaz.load_library()
lib.eaz_diag_persist_stamps.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros(256 * 32, np.uint64)
lib.eaz_diag_persist_stamps(search.h, buf.ctypes.data)
st = buf.reshape(256, 32).astype(np.int64)
st = st[st[:, 0] > 0]
w0, hp = st[:, :16], st[:, 16:]


def seg(name, a, b):
    d = b - a
    print(f"   {name:58s} median {np.median(d):8.0f}   p10 {np.percentile(d, 10):8.0f}   p90 {np.percentile(d, 90):8.0f} cycles")


print(f"{len(st)} workgroups, simulation {S // 2}; wave 0:")
seg("expand + backward", w0[:, 0], w0[:, 1])
seg("re-score (D+1 nodes, one lane each)", w0[:, 1], w0[:, 2])
seg("select walk + DeepSea step", w0[:, 2], w0[:, 3])
seg("W1 row gather of the wave's own trees + H1 store + barrier S1", w0[:, 3], w0[:, 4])
for h in range(3):
    seg(f"head {h}: layer 2 (256 MFMA per wave) + H2 round trip + quarter chain of layer 3", w0[:, 6 + 3 * h], w0[:, 7 + 3 * h])
    if h < 2:
        seg(f"head {h}: barrier", w0[:, 7 + 3 * h], w0[:, 6 + 3 * (h + 1)])
seg("head 2 done -> all heads' outputs visible (S7)", w0[:, 13], w0[:, 5])
seg("whole simulation (stamp 0 -> 5)", w0[:, 0], w0[:, 5])
