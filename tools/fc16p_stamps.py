#!/usr/bin/env python3
"""Phase breakdown of fc16p_forward_kernel (pipelined 16-row forward, all heads per workgroup) at the C3 shape from the diagnostic build
(make -C e-alphazero_amd/csrc diag): s_memtime stamps of thread 0 of every workgroup."""
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

W, B = 16, int(os.environ.get("B", 4096))
in_dim = (W + 32) * 5
env = engine.EnvHandle.subleq(W, (1,), True, device=0)
params = eaz.params.init_fc_params(in_dim, 256, W, seed=0)
net = engine.FCNet(env.ctx, in_dim, 256, W, params, eaz.params.init_hash_state(fill=0.5), hash_off=W * 5, hash_len=in_dim - W * 5)
states = env.init(B, key=0)
g = torch.Generator(device="cpu").manual_seed(0)
for k in range(6):
    states = env.step(states, torch.randint(0, W, (B,), generator=g, dtype=torch.int32).cuda())
obs = env.observe(states)
for _ in range(3):
    net.forward(obs)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    net.forward(obs)
b.record(); torch.cuda.synchronize()
print(f"net.forward (hash + 4 heads + ube_combine) {a.elapsed_time(b) / 50 * 1e3:.1f} us per call")
lib = eaz.load_library()
lib.eaz_diag_fc_stamps.argtypes = [C.c_void_p]
buf = np.zeros(1024 * 8, np.uint64)
lib.eaz_diag_fc_stamps(buf.ctypes.data)
st = buf.reshape(1024, 8)[: min(1024, (B + 15) // 16)].astype(np.int64)  # blockIdx.x < 1024: the last head launched wins the slot
names = ["observation tile -> f32 LDS tile, constants, barrier", "layer 1 (64 k-steps)", "H1 store + barrier",
         "layer 2 (64 k-steps)", "H2 store, quarter chain, barrier", "emit"]
d = np.diff(st[:, :7], axis=1)
for i, n in enumerate(names):
    print(f"   {n:62s} median {np.median(d[:, i]):8.0f}   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f} cycles")
print(f"   one workgroup (one head) median {np.median(st[:, 6] - st[:, 0]):.0f} cycles; start spread {np.ptp(st[:, 0])}")
