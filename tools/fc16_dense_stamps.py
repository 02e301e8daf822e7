#!/usr/bin/env python3
"""Phase breakdown of fc16_forward_kernel<0> (dense uint8 observations) at the C3 shape (Subleq-16 observations [48,5] = 240
inputs, B = 4096, all four heads) from the diagnostic build."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EAZ_LIB", os.path.join(ROOT, "e-alphazero_amd", "libeaz_diag.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import e_alphazero_amd as eaz
from e_alphazero_amd import engine
W, B = 16, 4096
in_dim = (W + 32) * 5
params = eaz.params.init_fc_params(in_dim, 256, W, seed=0)
env = engine.EnvHandle.subleq(W, (1,), True, device=0)
net = engine.FCNet(env.ctx, in_dim, 256, W, params, eaz.params.init_hash_state(fill=0.5), hash_off=W * 5, hash_len=in_dim - W * 5)
states = env.init(B, key=0)
g = torch.Generator(device="cpu").manual_seed(0)
for _ in range(6):
    states = env.step(states, torch.randint(0, W, (B,), generator=g, dtype=torch.int32).cuda())
obs = env.observe(states)
for _ in range(5): net.forward(obs)
torch.cuda.synchronize()
lib = eaz.load_library()
lib.eaz_diag_fc_stamps.argtypes = [C.c_void_p]
buf = np.zeros(1024 * 8, np.uint64)
lib.eaz_diag_fc_stamps(buf.ctypes.data)
st = buf.reshape(1024, 8).astype(np.int64)
st = st[st[:, 3] > st[:, 0]]
d = np.diff(st[:, :4], axis=1)
print(f"{len(st)} workgroups (4 heads x 256 row tiles)")
for i, n in enumerate(["stage + layer 1 (dense, 60 k-steps) + H1 + barrier", "layer 2 (dense MFMA)", "store H2 + barrier"]):
    print(f"   {n:52s} median {np.median(d[:, i]):8.0f}   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f} cycles")
print("   ideal MFMA cycles with 4 co-resident workgroups: layer 1", 60 * 4 * 32 * 4, " layer 2", 64 * 4 * 32 * 4)
