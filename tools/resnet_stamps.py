#!/usr/bin/env python3
"""Phase breakdown of resnet_forward_kernel (diagnostic build, make -C e-alphazero_amd/csrc diag): s_memtime stamps of the first
1024 workgroups of the LAST launch at B = 8192."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EAZ_LIB", os.path.join(ROOT, "e-alphazero_amd", "libeaz_diag.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import e_alphazero_amd as eaz
from e_alphazero_amd import engine
B, nb = 8192, 5
params, state = eaz.params.init_resnet_params(6, 7, 2, 7, 64, nb, seed=0)
env = engine.EnvHandle.connect4(device=0)
net = engine.ResNet(env.ctx, 6, 7, 2, 7, params, state, num_blocks=nb)
obs = env.observe(env.init(B, key=0))
for _ in range(3): net.forward(obs)
torch.cuda.synchronize()
lib = eaz.load_library()
lib.eaz_diag_fc_stamps.argtypes = [C.c_void_p]
buf = np.zeros(1024 * 8, np.uint64)
lib.eaz_diag_fc_stamps(buf.ctypes.data)
st = buf.reshape(1024, 8).astype(np.int64)
st = st[st[:, 5] > st[:, 0]]
d = np.diff(st[:, :6], axis=1)
tot = st[:, 5] - st[:, 0]
print(f"{len(st)} workgroups, median total {np.median(tot):.0f} cycles")
for i, n in enumerate(["input stage + conv0", "tower (5 blocks, 10 convs)", "head 1x1 convs + bn", "head linear layers", "final fc + tanh"]):
    print(f"   {n:28s} median {np.median(d[:, i]):9.0f}  ({100 * np.median(d[:, i]) / np.median(tot):4.1f} %)")
print(f"   ideal MFMA cycles of the tower per SIMD (8 samples, two waves of 11 and 10 tiles share a SIMD): {10 * 144 * 21 * 32}")
