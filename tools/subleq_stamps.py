#!/usr/bin/env python3
"""Phase breakdown of subleq16_search_kernel (one launch per SEARCH: EAZ_TUNE_PERSIST = 2, the round-2 production path; round 3's is
subleq16_selfplay_kernel, tools/sqp_stamps.py) at C3 from the diagnostic build (make -C e-alphazero_amd/csrc diag): s_memtime stamps of
simulation S/2, lane 0 of each of the four waves of every workgroup.  A stamp slot that was never written stays 0: differences are taken
only between slots that were both written (a wave that claimed no second environment has no stamp for it), never against a zero."""
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

W, B, S = 16, int(os.environ.get("B", 4096)), 64
cols, rows = 5, W + 32
in_dim, hash_off = rows * cols, W * cols
params = eaz.params.init_fc_params(in_dim, 256, W, seed=0)
env = engine.EnvHandle.subleq(W, (1,), True, device=0)
net = engine.FCNet(env.ctx, in_dim, 256, W, params, eaz.params.init_hash_state(fill=0.5), hash_off=hash_off, hash_len=in_dim - hash_off)
engine.tune(persist=2)
search = engine.Search(env, net, B, S, discount=0.97)
states = env.init(B, key=0)
search.selfplay(states, 6, key=0, use_graph=False)   # mid-episode states
obs = env.observe(states)
ex, xp, v, u, _ = net.forward(obs)
g = torch.zeros(B, W, device="cuda")
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
st = buf.reshape(256, 4, 8).astype(np.int64)  # waves 0-3 of the 8 (stamp slots of waves 4-7 alias lower ones: not recorded)
st = st[st[:, 0, 0] > 0]


def seg(name, d):
    d = np.asarray(d).reshape(-1)
    if d.size == 0:
        print(f"   {name:58s} (no wave wrote both stamps)")
        return
    assert (d >= 0).all(), f"{name}: negative interval -- a stamp pair was read out of order"
    print(f"   {name:58s} median {np.median(d):8.0f}   p10 {np.percentile(d, 10):8.0f}   p90 {np.percentile(d, 90):8.0f} cycles   ({d.size} samples)")


def span(a, b):
    """b - a where both stamps were written by THIS launch (the buffer is not cleared between launches: a slot the wave did not reach
    this time holds an older, i.e. smaller, stamp or 0)"""
    ok = (a > 0) & (b > a)
    return (b - a)[ok]


w0 = st[:, 0, :]
print(f"{len(st)} workgroups, simulation {S // 2} (100 MHz s_memtime ticks x clock ratio: see persist_stamps.py)")
seg("T: tree step (wave 0)", w0[:, 1] - w0[:, 0])
seg("T -> E barrier (wave 0)", w0[:, 2] - w0[:, 1])
seg("E: 4 interpreters in turn (wave 0)", w0[:, 3] - w0[:, 2])
seg("E: slowest wave of the workgroup", (st[:, :, 3] - st[:, :, 2]).max(axis=1))
seg("E -> F barrier (wave 0)", w0[:, 4] - w0[:, 3])
seg("F: 3 heads (wave 0)", w0[:, 5] - w0[:, 4])
seg("F -> next barrier", w0[:, 6] - w0[:, 5])
seg("one simulation", w0[:, 6] - w0[:, 0])

lib.eaz_diag_sq_stamps.argtypes = [C.c_void_p]
q = np.zeros(16, np.uint64)
lib.eaz_diag_sq_stamps(q.ctypes.data)
q = q.astype(np.int64)
n = max(int(q[15]), 1)
print(f"inside subleq16_wave_body, mean over the {n} calls of workgroup 0's first wave (all launches of this process):")
for name, i in (("loads of the parent state", 0), ("decision + program patch", 1), ("test cases + interpreter", 2), ("assemble + state stores", 3),
                ("row-words / observation row", 4), ("hash (per-simulation kernels only)", 5)):
    print(f"   {name:40s} {q[i] / n:8.0f} cycles")
print(f"   interpreter runs per call {q[9] / n:.2f}, executed cycles per run {q[8] / max(int(q[9]), 1):.1f} (longest run seen: {q[10]})")

# tree_step_body's own stamps (tree.hpp EAZ_TSTAMP): the last call with a select phase = simulation S-1, thread 0 of each workgroup
lib.eaz_diag_tree_stamps.argtypes = [C.c_void_p]
tb = np.zeros(1024 * 8, np.uint64)
lib.eaz_diag_tree_stamps(tb.ctypes.data)
tb = tb.reshape(1024, 8).astype(np.int64)[:256]
print(f"tree_step_body, simulation {S - 1} (select depth of tree 0 of the workgroup: median {np.median(tb[:, 5]):.0f}):")
seg("expand: loads + new node stores", tb[:, 1] - tb[:, 0])
seg("backward", tb[:, 2] - tb[:, 1])
seg("select walk", tb[:, 3] - tb[:, 2])
seg("result stores", tb[:, 4] - tb[:, 3])

# interpreter phase per wave (8 waves x [start, after its first environment, after its second])
lib.eaz_diag_persist_stamps2.argtypes = [C.c_void_p, C.c_void_p]
eb = np.zeros(256 * 32, np.uint64)
lib.eaz_diag_persist_stamps2(search.h, eb.ctypes.data)
eb = eb.reshape(256, 8, 4).astype(np.int64)
eb = eb[eb[:, 0, 0] > 0]
print("interpreter phase, all 8 waves of every workgroup:")
# slots: 0 = phase start, 1 / 2 / 3 = after the wave's first / second / third-or-later environment; a wave that got no further environment
# from the work queue leaves the later slots at 0
e0 = eb[:, :, 0]
last = np.where(eb[:, :, 3] > e0, eb[:, :, 3], np.where(eb[:, :, 2] > e0, eb[:, :, 2], eb[:, :, 1]))   # the wave's last stamp of this launch
seg("first environment of a wave", span(eb[:, :, 0], eb[:, :, 1]))
seg("second environment of a wave (waves that claimed one)", span(eb[:, :, 1], eb[:, :, 2]))
busy = np.where((e0 > 0) & (last > e0), last - e0, 0)
seg("slowest wave of the workgroup (all its environments)", busy.max(axis=1))
seg("fastest wave of the workgroup (all its environments)", np.where(busy > 0, busy, np.iinfo(np.int64).max).min(axis=1))
first = np.where(eb[:, :, 1] > e0, eb[:, :, 1] - e0, 0)
second = np.where(eb[:, :, 2] > eb[:, :, 1], eb[:, :, 2] - eb[:, :, 1], 0)
seg("slowest single environment of the workgroup (first two per wave)", np.maximum(first, second).max(axis=1))

# fc16p_body's own stamps (net_fc.hpp EAZ_STAMP), thread 0 of each workgroup: its last call = the head that runs alone in round 2
lib.eaz_diag_fc_stamps.argtypes = [C.c_void_p]
fb = np.zeros(1024 * 8, np.uint64)
lib.eaz_diag_fc_stamps(fb.ctypes.data)
fb = fb.reshape(1024, 8).astype(np.int64)[:256]
print("fc16p_body, last head of the last simulation (runs alone on its four waves):")
for name, i in (("constants + weight prologue -> tile barrier", 0), ("layer 1", 1), ("H1 store + barrier", 2), ("layer 2", 3),
                ("H2 store + quarter chain + barrier", 4), ("emit", 5)):
    seg(name, fb[:, i + 1] - fb[:, i])
