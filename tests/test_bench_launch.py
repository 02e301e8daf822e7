"""bench.py's N > 1 entry: `python bench.py --gpus N` (the driver's form) must start N ranks itself when it is not already
running under torchrun, before anything touches the GPU, and rank 0 must report n_gpus = N.  Runs on CPU through --dry-run
(launch path + gloo process group + the trajectory all-gather; no device work, `value` null)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected exactly one JSON line, got {len(lines)}:\n{stdout}"
    return json.loads(lines[0])


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n", [1, 2])
def test_self_launch_reports_n_gpus(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "0", "--dry-run", "--batch", "8",
                        "--selfplay-steps", "3"], env=_clean_env(), capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _json_line(r.stdout)
    assert res["n_gpus"] == n and res["steps"] == 2 and res["value"] is None and res["pooled_block_ok"] is True
    assert res["config"]["global_batch"] == 8 * n and res["scaling"] == "weak"


@pytest.mark.timeout(300)
def test_eight_rank_dry_run_with_the_c5_record_layout():
    """BASELINE.json configs[4] (DeepSea-32, 8 x B, RCCL all-gather) rehearsed on the CPU at a reduced per-rank batch: eight ranks started
    by bench.py itself, C2/C5 record layout (16-byte states, 2 actions, T = 32), three steps through the asynchronous double-buffered
    pooler, every rank's block checked at its place in the [8 * B, T, R] replay block (src/main.py:383-385)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "3", "--warmup", "0", "--dry-run", "--batch", "16", "--workload", "c2"],
                       env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _json_line(r.stdout)
    assert res["n_gpus"] == 8 and res["pooled_block_ok"] is True and res["config"]["global_batch"] == 128
    assert "x8" in res["config"]["parallelism"]


@pytest.mark.timeout(180)
def test_torchrun_environment_is_honoured():
    """Under torchrun (WORLD_SIZE set) bench.py must not fan out again, and --gpus must agree with the world size."""
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--dry-run", "--batch", "4"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--dry-run", "--batch", "4"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


@pytest.mark.timeout(180)
def test_failed_rank_fails_the_launch():
    """A rank that dies must take the launch down with a non-zero exit code instead of leaving the others in a collective."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--dry-run", "--batch", "-1"], env=_clean_env(),
                       capture_output=True, text=True, timeout=170)
    assert r.returncode != 0


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("ranks,batch", [(2, 256), (5, 128)])
def test_ranks_on_the_gpu_box_through_the_self_launch(ranks, batch):
    """The real engine under the N > 1 entry: `python bench.py --gpus N` on a box with fewer GPUs than ranks is a rehearsal (the ranks
    share the card, gloo all-gather through host memory) -- every line of the sharded path runs except RCCL itself: C5's record layout
    (DeepSea-32, 44-byte records), the asynchronous double-buffered pooler, the barrier / max-over-ranks timing, the per-rank times.
    `n_gpus` is N, the value is not a scaling number and says so.  Five ranks is what a one-GPU box of this pool allows beside the test
    process itself (at most six processes on the card); the eight-rank case needs the node and runs under the driver only."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--batch", str(batch), "--selfplay-steps", "4",
                        "--no-cpu-baseline", "--no-extra"], env=_clean_env(), capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-3000:]
    res = _json_line(r.stdout)
    assert res["n_gpus"] == ranks and res["config"]["global_batch"] == ranks * batch and res["value"] > 0
    by_rank = res["ms_per_step_by_rank"]
    assert len(by_rank["ranks"]) == ranks and 0 < by_rank["min"] <= by_rank["max"]
    import torch

    if torch.cuda.device_count() < ranks:
        assert "rehearsal" in res
