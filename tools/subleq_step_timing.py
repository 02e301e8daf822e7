"""Per-cycle cost of subleq_step_kernel (the per-lane interpreter of every word size other than 16): env.step on batches whose programs are
known -- all-zero memory (period 1 from the first cycle: the floor of a launch) and a counting loop that never repeats a memory state
(200 interpreted cycles of test case 0) -- timed with events around the launch.  python tools/subleq_step_timing.py [W] [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e_alphazero_amd as eaz  # noqa: E402
from e_alphazero_amd import engine  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
env = engine.EnvHandle.subleq(W, (1,), True, device=0)


def timed(st, a, reps=5):
    act = torch.full((B,), a, dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = env.step(st.clone(), act, key=1, autoreset=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return out, float(np.median(ts))


st = env.init(B, key=3)
_, t_zero = timed(st, 0)
print(f"W={W} B={B}: all-zero program (period 1 at once): {t_zero:.1f} us per launch (clone + step)")
# counting loop: word 0..2 = (10, 11, 0), word 11 = 1: mem[10] -= 1 each visit; nothing repeats within 200 cycles
prog = [10, 11, 0] + [0] * 8 + [1]
for w in prog:
    st, t = timed(st, w, reps=1)
_, t_cnt = timed(st, 0)
print(f"counting loop (12 words written): {t_cnt:.1f} us per launch -> {(t_cnt - t_zero) * 2.4e3 / 200:.0f} shader clocks per Subleq cycle if one test case runs 200 cycles")
