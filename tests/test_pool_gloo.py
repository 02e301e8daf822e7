"""N > 1 path on CPU: two gloo ranks pool their [T, B, ...] trajectory blocks with one all-gather (e-alphazero_amd/pool.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_output(rank, T, B, SB, A, step=0):
    g = torch.Generator().manual_seed(100 + rank + 1000 * step)
    return dict(
        state=torch.randint(0, 256, (T, B, SB), dtype=torch.uint8, generator=g),
        root_value=torch.randn(T, B, generator=g), root_epistemic_std=torch.rand(T, B, generator=g),
        value_prediction=torch.randn(T, B, generator=g), ube_prediction=torch.rand(T, B, generator=g),
        q_values_epistemic_variance=torch.rand(T, B, A, generator=g),
        action=torch.randint(0, A, (T, B), dtype=torch.int32, generator=g),
    )


def _worker(rank, world, port, T, B, SB, A, q):
    sys.path.insert(0, ROOT)
    import e_alphazero_amd  # noqa: F401
    from e_alphazero_amd import pool

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _fake_output(rank, T, B, SB, A)
    block = pool.pack(out)
    assert block.shape == (T, B, pool.record_bytes(SB, A))
    gathered = pool.all_gather_trajectories(block)
    pooled = pool.unpack(pool.pooled_view(gathered), SB, A)
    ok = True
    for r in range(world):
        ref = _fake_output(r, T, B, SB, A)
        for k, v in ref.items():
            ok &= bool(torch.equal(pooled[k][:, r * B:(r + 1) * B], v))
    # replay landing layout (src/main.py:383-385): [world * B, T, ...], rank-major along the batch axis
    bm = pool.unpack_batch_major(pool.pooled_batch_major(gathered), SB, A)
    for r in range(world):
        ref = _fake_output(r, T, B, SB, A)
        for k, v in ref.items():
            ok &= bool(torch.equal(bm[k][r * B:(r + 1) * B], v.transpose(0, 1)))
    ok &= tuple(bm["state"].shape) == (world * B, T, SB) and tuple(bm["action"].shape) == (world * B, T)
    # three consecutive selfplay() calls through the asynchronous, double-buffered pooler (what bench.py's timed loop does): different data
    # every step, results read late (after the NEXT submit), buffers reused from the third step on
    pooler = pool.TrajectoryPooler()
    pend = []
    for step in range(3):
        pend.append(pooler.submit(_fake_output(rank, T, B, SB, A, step)))
        if step >= 1:  # step - 1 is complete and still in its landing buffer while step's collective is in flight
            got = pool.unpack_batch_major(pool.pooled_batch_major(pend[step - 1].wait()), SB, A)
            for r in range(world):
                ref = _fake_output(r, T, B, SB, A, step - 1)
                for k, v in ref.items():
                    ok &= bool(torch.equal(got[k][r * B:(r + 1) * B], v.transpose(0, 1)))
    pooler.flush()
    got = pool.unpack_batch_major(pool.pooled_batch_major(pend[2].wait()), SB, A)
    for r in range(world):
        for k, v in _fake_output(r, T, B, SB, A, 2).items():
            ok &= bool(torch.equal(got[k][r * B:(r + 1) * B], v.transpose(0, 1)))
    ok &= pend[0].gathered.data_ptr() == pend[2].gathered.data_ptr() != pend[1].gathered.data_ptr()  # two landing buffers, alternating
    try:  # ... so the handle of step 0 is stale once step 2 has been submitted, and says so instead of returning step 2's data
        pend[0].wait()
        ok = False
    except RuntimeError as e:
        ok &= "overwritten" in str(e)
    q.put((rank, ok, tuple(pooled["state"].shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_pack_unpack_roundtrip():
    sys.path.insert(0, ROOT)
    import e_alphazero_amd  # noqa: F401
    from e_alphazero_amd import pool

    out = _fake_output(0, 3, 5, 16, 2)
    back = pool.unpack(pool.pack(out), 16, 2)
    assert all(torch.equal(back[k], out[k]) for k in out)


@pytest.mark.timeout(120)
def test_two_rank_all_gather_gloo():
    world, T, B, SB, A = 2, 4, 6, 16, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, B, SB, A, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(shape == (T, world * B, SB) for _, _, shape in res)
