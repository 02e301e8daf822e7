#!/usr/bin/env python3
"""Phase breakdown of the 64-row fc_forward_kernel from the diagnostic build (make -C e-alphazero_amd/csrc diag; EAZ_LIB=.../libeaz_diag.so).
Reads the in-kernel s_memtime stamps of the LAST launch and prints the median share of each phase over the workgroups."""
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

G, B = 32, 4096
engine.tune(fc_tile=64)  # this tool stamps the 64-row kernel (fc_forward_kernel); fc16_stamps.py / fc16_dense_stamps.py do the 16-row one
am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
params = eaz.params.init_fc_params(G * G, 256, 2, seed=0)
env = engine.EnvHandle.deepsea(G, am, device=0)
net = engine.FCNet(env.ctx, G * G, 256, 2, params, eaz.params.init_hash_state(fill=0.5))
lib = eaz.load_library()
lib.eaz_diag_fc_stamps.argtypes = [C.c_void_p]
names = ["stage obs -> LDS", "layer 1 (sparse MFMA)", "store H1 + barrier", "layer 2 (dense MFMA)", "store H2 + barrier", "layer 3 + epilogue"]
rng = np.random.default_rng(1)
cases = {"identical one-hot": None, "random one-hot": None, "dense 30%": None}
for name in cases:
    obs = np.zeros((B, G * G), np.uint8)
    if name == "identical one-hot":
        obs[:, 37] = 1
    elif name == "random one-hot":
        obs[np.arange(B), rng.integers(0, G * G, B)] = 1
    else:
        obs = (rng.random((B, G * G)) < 0.3).astype(np.uint8)
    o = torch.from_numpy(obs).cuda()
    for _ in range(5):
        net.forward(o)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, np.uint64)
    lib.eaz_diag_fc_stamps(buf.ctypes.data)
    st = buf.reshape(1024, 8)[:256].astype(np.int64)
    st = st[st[:, 6] > st[:, 0]]
    d = np.diff(st[:, :7], axis=1)
    tot = st[:, 6] - st[:, 0]
    print(f"{name}: {len(st)} workgroups, median total {np.median(tot):.0f} cycles")
    for i, n in enumerate(names):
        print(f"   {n:24s} {np.median(d[:, i]):9.0f} cycles  ({100 * np.median(d[:, i]) / np.median(tot):4.1f} %)")
