#!/usr/bin/env python3
"""Phase breakdown of subleq16_selfplay_kernel at C3 from the diagnostic build (make -C e-alphazero_amd/csrc diag): s_memtime stamps of
simulation S/2 of the first real step, lane 0 of each of the eight waves of every workgroup (search_subleq_persist.hpp EAZ_QPSTAMP).
Stamp slots that a wave never wrote stay 0 and are dropped, never subtracted."""
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

W, B, S, T = 16, int(os.environ.get("B", 4096)), 64, int(os.environ.get("T", 8))
BETA = float(os.environ.get("BETA", 0.0))
cols, rows = 5, W + 32
in_dim, hash_off = rows * cols, W * cols
params = eaz.params.init_fc_params(in_dim, 256, W, seed=0)
env = engine.EnvHandle.subleq(W, (1,), True, device=0)
ORDER = os.environ.get("ORDER", "constants-first")   # layer-1 input order (bench.py --l1-order): constants-first = 5 half-rounds of layer 1, ascending = 7
order = eaz.params.subleq_constants_first(W, True) if ORDER == "constants-first" else None
net = engine.FCNet(env.ctx, in_dim, 256, W, params, eaz.params.init_hash_state(fill=0.5), hash_off=hash_off, hash_len=in_dim - hash_off, input_order=order)
search = engine.Search(env, net, B, S, discount=0.97, exploration=BETA != 0.0)
states = env.init(B, key=0)
kw = dict(directed_exploration=True, exploration_beta=BETA) if BETA else {}
search.selfplay(states, 6, key=0, **kw)   # mid-episode states
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    search.selfplay(states.clone(), T, key=1 + rep, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"selfplay(T={T}): {1e3 * dt:.3f} ms = {1e6 * dt / (T * (S + 1)):.2f} us per forward (host-timed, one launch), {B * T / dt:.0f} env-steps/s")
lib = eaz.load_library()
buf = np.zeros(256 * 8 * 16, np.uint64)
lib.eaz_diag_qp_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.eaz_diag_qp_stamps(search.h, buf.ctypes.data)
st = buf.reshape(256, 8, 16).astype(np.int64)
st = st[(st[:, :, 4] > 0).all(axis=1)]


def seg(name, d):
    d = np.asarray(d).reshape(-1)
    print(f"   {name:62s} median {np.median(d):8.0f}   p10 {np.percentile(d, 10):8.0f}   p90 {np.percentile(d, 90):8.0f} cycles")


def diff(a, b):
    """stamps[..., b] - stamps[..., a] over the waves that wrote both"""
    x, y = st[:, :, a], st[:, :, b]
    ok = (x > 0) & (y > 0)
    return (y - x)[ok]


print(f"{len(st)} workgroups, simulation {S // 2} of the first real step (shader-clock ticks of s_memtime)")
seg("T: expand + backward (every wave: its two trees)", diff(0, 1))
seg("T: re-score of the path's nodes", diff(1, 2))
seg("T: select walk", diff(2, 3))
seg("T -> E barrier", diff(3, 4))
seg("E: interpreters of one wave (work queue)", diff(4, 5))
seg("E: slowest wave of the workgroup", (st[:, :, 5] - st[:, :, 4]).max(axis=1))
seg("E -> F barrier (wait for the slowest wave)", diff(5, 6))
seg("F: observation tile + three heads in turn", diff(6, 7))
# F by XCD (workgroup b runs on XCD b % 8 when the grid is 256 workgroups on 256 CUs) and by position inside the XCD: is the spread a property of
# the place (L2 slice, clocks) or of the moment?
if len(st) == 256:
    print("   F by wave (median of stamp 6 -> 7): " + "  ".join(f"w{w}: {np.median(st[:, w, 7] - st[:, w, 6]):.0f}" for w in range(8)))
    print("   arrival at stamp 7 relative to the workgroup's first wave there (median): " +
          "  ".join(f"w{w}: {np.median(st[:, w, 7] - st[:, :, 7].min(axis=1)):.0f}" for w in range(8)))
    fw0 = (st[:, 0, 7] - st[:, 0, 6])
    print("   F of wave 0 by XCD (median): " + "  ".join(f"{x}: {np.median(fw0[x::8]):.0f}" for x in range(8)))
    starts = st[:, 0, 6] - st[:, 0, 6].min()
    print(f"   start of F relative to the earliest workgroup: median {np.median(starts):.0f}, p90 {np.percentile(starts, 90):.0f}; "
          f"correlation of F's duration with its start {np.corrcoef(starts, fw0)[0, 1]:.2f}")
lib.eaz_diag_qp_fwd.argtypes = [C.c_void_p]
fwb = np.zeros(2048 * 8, np.uint64)
lib.eaz_diag_qp_fwd(fwb.ctypes.data)
fw = fwb.reshape(256, 8, 8).astype(np.int64)[:, 0, :]
if (fw[:, :7] > 0).all():
    print("   inside F, wave 0 (median): " + "  ".join(
        f"head {h}: layer 1 {np.median(fw[:, 1 + 2 * h] - (fw[:, 2 * h] if h == 0 else fw[:, 2 * h])):.0f}, layer 2 (+ H1 tile, barrier) {np.median(fw[:, 2 + 2 * h] - fw[:, 1 + 2 * h]):.0f};"
        for h in range(3)) + f"  after the last layer 2 -> stamp 7: {np.median(st[:, 0, 7] - fw[:, 6]):.0f}")
lib.eaz_diag_qp_rounds.argtypes = [C.c_void_p]
rb = np.zeros(2048 * 64, np.uint64)
lib.eaz_diag_qp_rounds(rb.ctypes.data)
rr = rb.reshape(256, 8, 64).astype(np.int64)[:, 0, :]
N1 = 5 if order else 7  # half-rounds of layer 1 (sq16_l1_halves); stamps sit behind PAIRS of half-rounds: slots head * 16 + 1, 3, .., N1 - 2, N1 - 1
if (fw[:, :7] > 0).all():   # (layer 1) and head * 16 + 9, 11, 13, 15 (layer 2); a stamp per half-round cost ~100 clocks each in the diagnostic build
    for h in range(3):
        prev = fw[:, 2 * h]  # F start / end of layer 2 of the head before
        l1_slots = [1] + list(range(3, N1 - 2, 2)) + [N1 - 2, N1 - 1]
        l1, last = [], prev
        for sl in l1_slots:
            l1.append(np.median(rr[:, h * 16 + sl] - last))
            last = rr[:, h * 16 + sl]
        l2, last = [], fw[:, 1 + 2 * h]
        for sl in (9, 11, 13, 15):
            l2.append(np.median(rr[:, h * 16 + sl] - last))
            last = rr[:, h * 16 + sl]
        print(f"   head {h}, wave 0, clocks (floor 1024 per half-round of 8 k-steps): layer 1 by stamps behind half-rounds {l1_slots}: " + " ".join(f"{x:.0f}" for x in l1) +
              "   | H1 tile + barrier + layer 2 by pairs: " + " ".join(f"{x:.0f}" for x in l2))
t0 = st[:, :, 0].min(axis=1)
seg("T (first wave in -> last wave at the T/E barrier)", st[:, :, 4].max(axis=1) - t0)
seg("E (barrier to barrier)", st[:, :, 6].max(axis=1) - st[:, :, 4].max(axis=1))
seg("one simulation (T start -> F end, wave 0)", st[:, 0, 7] - st[:, 0, 0])

print("inside the FIRST interpreter step of every wave (qp_env_step, raw stamps):")
seg("   compact state from LDS, decision, test vectors requested", diff(8, 9))
seg("   program patch + test cases + interpreter", diff(9, 10))
seg("   reward / flags", diff(10, 11))
seg("   new state to LDS + observation words / bit masks", diff(11, 12))
seg("   (state blob: real steps only)", diff(12, 13))
seg("   outputs + stamp", diff(13, 14))
seg("   the whole step", diff(8, 14))

lib.eaz_diag_qp_dhist.argtypes = [C.c_void_p]
dh = np.zeros(72, np.uint64)
lib.eaz_diag_qp_dhist(dh.ctypes.data)
dh = dh.astype(np.int64)
tot = max(int(dh[:64].sum()), 1)
print("select depth D over the first real steps of 64 workgroups (share of tree-simulations): " +
      "  ".join(f"D={d}: {100 * dh[d] / tot:.1f}%" for d in range(1, 16) if dh[d]))
print("mean D by eighth of the search: " + "  ".join(f"{8 * dh[64 + i] / tot:.2f}" for i in range(8)))

lib.eaz_diag_qp_ehist.argtypes = [C.c_void_p, C.c_int]
eh = np.zeros(128, np.uint64)
lib.eaz_diag_qp_ehist(eh.ctypes.data, 0)
eh = eh.astype(np.int64)
n = max(int(eh[0:4].sum()), 1)
print(f"interpreter steps of the leaf env.step (64 workgroups, every simulation of the calls above): {n}; no run needed (the parent's runs never "
      f"touched the new word): {100 * eh[0] / n:.1f}%; test cases run: " + "  ".join(f"{c}: {100 * eh[c] / n:.1f}%" for c in (1, 2, 3))
      + f";  mean interpreted cycles per step {eh[100] / n:.1f}")


def hist(name, h):
    tot = max(int(h.sum()), 1)
    cum = np.cumsum(h) / tot
    print(f"   {name:48s} " + "  ".join(f"<{4 * (i + 1)}: {100 * cum[i]:.1f}%" for i in (0, 1, 2, 3, 5, 7, 11, 15, 23, 30)) + f"  >=124: {100 * h[31] / tot:.2f}%")


hist("interpreted cycles of a whole step (all cases)", eh[4:36])
hist("of test case 0 alone", eh[36:68])
hist("of the steps that ran more than one case", eh[68:100])

# cost model of one interpreter step: clocks between stamps 9 and 10 (program patch + test cases + interpreter) against the interpreted cycles
x, y0, y1 = st[:, :, 15], st[:, :, 9], st[:, :, 10]
ok = (x > 0) & (y0 > 0) & (y1 > 0)
cyc, clk = (x[ok] - 1).astype(np.float64), (y1 - y0)[ok].astype(np.float64)
if len(cyc) > 8:
    A = np.stack([np.ones_like(cyc), cyc], 1)
    coef, *_ = np.linalg.lstsq(A, clk, rcond=None)
    print(f"interpreter segment of the stamped steps ({len(cyc)} samples): clocks = {coef[0]:.0f} + {coef[1]:.0f} x interpreted cycles (least squares); "
          f"mean cycles {cyc.mean():.1f}, max {cyc.max():.0f}")
    for lo, hi in ((0, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 1000)):
        m = (cyc >= lo) & (cyc < hi)
        if m.any():
            print(f"   cycles in [{lo}, {hi}): {m.sum():5d} steps, median {np.median(clk[m]):7.0f} clocks")

lib.eaz_diag_qp_fine.argtypes = [C.c_void_p]
fb = np.zeros(2048 * 8, np.uint64)
lib.eaz_diag_qp_fine(fb.ctypes.data)
fine = fb.reshape(256, 8, 8).astype(np.int64)
raw = buf.reshape(256, 8, 16).astype(np.int64)
okf = (fine[:, :, 0] > 0) & (fine[:, :, 1] > 0) & (fine[:, :, 2] > 0) & (raw[:, :, 9] > 0) & (raw[:, :, 10] > 0) & (raw[:, :, 15] > 0)
short = okf & (raw[:, :, 15] - 1 <= 1)
print(f"inside the interpreter segment, steps of <= 1 interpreted cycle ({short.sum()} samples):")
for name, d in (("stamp 9 -> program patched", fine[:, :, 0] - raw[:, :, 9]), ("-> test vectors of case 0 in registers", fine[:, :, 1] - fine[:, :, 0]),
                ("-> subleq16_wave_run returned", fine[:, :, 2] - fine[:, :, 1]), ("-> stamp 10 (loop exit)", raw[:, :, 10] - fine[:, :, 2])):
    seg("   " + name, d[short])

# interpreter work queue: what a better order could buy.  Cost model of one step from the regression above; 8 waves take the environments in
# queue order (list scheduling), the phase ends with the last wave.
lib.eaz_diag_qp_its.argtypes = [C.c_void_p]
its = np.zeros(64 * 64 * 16, np.uint32)
lib.eaz_diag_qp_its(its.ctypes.data)
np.save(os.path.join(ROOT, "gpurun_out", "qp_its.npy"), its)
its = its.reshape(64, 64, 16)
okq = ((its >> 16) & 1).all(axis=2)
cost, pcost = (its & 0xFF).astype(np.float64), ((its >> 8) & 0xFF).astype(np.float64)
fixed, per = 4000.0, 423.0


def wall(c, order):
    t = np.zeros(8)
    for j in order:
        k = int(np.argmin(t))
        t[k] += fixed + per * c[j]
    return t.max()


res = {"fifo": [], "lpt_true": [], "lpt_parent": [], "ideal": []}
for w in range(64):
    for sim in range(16, 64):
        if not okq[w, sim]:
            continue
        c, pc = cost[w, sim], pcost[w, sim]
        res["fifo"].append(wall(c, range(16)))
        res["lpt_true"].append(wall(c, np.argsort(-c, kind="stable")))
        res["lpt_parent"].append(wall(c, np.argsort(-pc, kind="stable")))
        res["ideal"].append(max((16 * fixed + per * c.sum()) / 8, fixed + per * c.max()))
if res["fifo"]:
    print("modelled interpreter phase (clocks, simulations 16..63 of the first real step, 64 workgroups): " +
          "  ".join(f"{k}: mean {np.mean(v):.0f}" for k, v in res.items()))
    cc = np.corrcoef(cost[okq].reshape(-1), pcost[okq].reshape(-1))[0, 1]
    print(f"correlation of a step's interpreted cycles with its parent's: {cc:.2f}")

