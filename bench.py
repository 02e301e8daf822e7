#!/usr/bin/env python3
"""bench.py -- self-play env-steps/sec of the HIP engine on BASELINE.json's headline configuration.

A "step" is one selfplay() call (src/selfplay.py:78-149): T = 32 real environment steps (config.py:37) of a batch of
B = 4096 DeepSea-32 environments per GPU, 64 Gumbel-MuZero simulations per real step, FC H=256 network -- BASELINE.json
configs[1] ("C2").  env-steps = B_total * T per step (src/main.py:378); search-internal steps are not counted.

N > 1: one process per GPU (torchrun), the batch is sharded (weak scaling, B per GPU fixed); the only collective is one RCCL
all-gather per selfplay() call that pools the [T, B, ...] trajectory block of every rank (the counterpart of
src/main.py:383-385), inside the timed region.

Prints ONE JSON line on rank 0 (see the task contract), with `roofline` for the dominant kernel (fused FC forward,
f32 MFMA) measured live with HIP events on the launch stream, `roofline_gather` for the leaf-gather kernel (HBM),
and `cpu_baseline` = the CPU oracle timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import e_alphazero_amd as eaz  # noqa: E402
from e_alphazero_amd import engine, pool  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBPS = 8000.0        # MI355X_MICROARCH.md "HBM3E peak BW"


def fc_flops_per_eval(in_dim, H, A, heads):
    f = 0
    for h in heads:
        out = 1 if h < 2 else A
        f += 2 * (in_dim * H + H * H + H * out)
    return f


def cpu_share():
    """Host cores this process may actually use: affinity mask, cgroup CPU quota, and the 16-core share of a 1-GPU box."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // p))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=32, help="DeepSea size")
    ap.add_argument("--batch", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--selfplay-steps", type=int, default=32)
    ap.add_argument("--hash-fill", type=float, default=0.5, help="fraction of the novelty bit-set that is set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=4096)
    ap.add_argument("--cpu-sample-steps", type=int, default=8)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    n_gpus = world
    dev = torch.device("cuda", local_rank)

    G, B, S, T, A, H = args.grid, args.batch, args.sims, args.selfplay_steps, 2, 256
    # synthetic inputs: action_map ~ Bernoulli(0.5) seed 0, random-init FC net seed 0, half-filled bit set
    am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
    params = eaz.params.init_fc_params(G * G, H, A, seed=0)
    hstate = eaz.params.init_hash_state(fill=args.hash_fill, seed=0)
    env = engine.EnvHandle.deepsea(G, am, device=local_rank)
    net = engine.FCNet(env.ctx, G * G, H, A, params, hstate)
    search = engine.Search(env, net, B, S, discount=0.997)
    states = env.init(B, key=rank)
    outputs = None
    gathered = None

    def one_step(i):
        nonlocal outputs, gathered
        outputs = search.selfplay(states, T, key=(1000 * rank + i), outputs=outputs)
        if dist is not None:
            # pool the trajectories of all ranks (replay-buffer landing zone): ONE all-gather of the packed block
            gathered = pool.all_gather_trajectories(pool.pack(outputs), out=gathered)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(-1 - i)
    # ---- timed region: EXACTLY K steps in the production configuration (one captured hipGraph per real step) ----
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    sync()
    dt = time.perf_counter() - t0
    # ---- instrumented pass: the same K steps again with HIP events recorded on the launch stream around every kernel
    # (event records cannot sit between the nodes of a replayed graph, so this pass uses plain stream launches; kernel
    # durations are unaffected, only the host-side launch gaps differ).  Feeds `roofline`, never `value`. ----
    search.profile(True)
    search.profile_read(reset=True)
    sync()
    t1 = time.perf_counter()
    for i in range(args.steps):
        outputs = search.selfplay(states, T, key=(1000 * rank + 500 + i), outputs=outputs, use_graph=False)
    sync()
    dt_instr = time.perf_counter() - t1
    prof = search.profile_read(reset=True)
    search.profile(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    env_steps = B * n_gpus * T * args.steps
    value = env_steps / dt
    result = {
        "metric": "self-play env-steps/sec (batch 4096, 64 sims)",
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded DeepSea action map, random-init FC net, half-filled novelty set)",
        "config": {"workload": f"DeepSea size={G}, batch={B}/GPU, {S} sims, {T} real steps per selfplay() call, FC H={H} (BASELINE.json configs[1])",
                   "global_batch": B * n_gpus, "parallelism": f"batch-sharded x{n_gpus}, RCCL all-gather of trajectories" if n_gpus > 1 else "single GPU"},
    }
    # ---- roofline of the dominant kernel (fused FC forward; 3 heads evaluated: value, ube, exploitation) -----------
    # `achieved` counts the FLOPs the kernel EXECUTED: layers 2-3 are dense; in layer 1 every 64x4 observation block that
    # is entirely zero is skipped (exact: fma(0,w,acc)=acc), so executed <= dense.  `dense_equiv` prices the same
    # launches at the dense count of SURVEY.md 8(d) (2 * MACs of every layer).
    heads = (0, 1, 2)
    n_launch = max(prof["net"]["launches"], 1)
    dense_flops = fc_flops_per_eval(G * G, H, A, heads) * B
    l23 = sum(2 * (H * H + H * (1 if h < 2 else A)) for h in heads) * B
    l1_exec = prof["net"]["layer1_ksteps"] * (64 * H * 4 * 2) / n_launch
    flops = l23 + l1_exec
    net_ms = prof["net"]["ms"] / n_launch
    ach = flops / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0
    result["roofline"] = {"kernel": "fc_forward_kernel", "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                          "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                          "avg_launch_us": 1e3 * net_ms, "launches": prof["net"]["launches"],
                          "flops_per_launch": flops, "dense_flops_per_launch": dense_flops,
                          "dense_equiv_tflops": dense_flops / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0}
    # leaf gather: algorithmic bytes on the engine's compact layout = parent state + child state + observation row
    gbytes = B * (2 * env.state_bytes + env.obs_dim)
    g_ms = prof["gather"]["ms"] / max(prof["gather"]["launches"], 1)
    gach = gbytes / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
    # HBM-side traffic per launch from the PMC passes committed under profiles/ (rocprofv3 FETCH_SIZE / WRITE_SIZE cannot be
    # collected from inside this process); only attached when the file was recorded for this workload
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r01_b_pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("workload") == f"DeepSea-{G}, B={B}, {S} sims":
            for kname, field in (("eaz::fc_forward_kernel", "roofline"), ("eaz::leaf_gather_deepsea_kernel", "roofline_gather")):
                k = pmc["kernels"].get(kname)
                if k:
                    traffic[field] = 1024.0 * (k.get("FETCH_SIZE_KB_per_launch", 0.0) + k.get("WRITE_SIZE_KB_per_launch", 0.0))
    except (OSError, ValueError):
        pass
    if "roofline" in traffic:
        result["roofline"]["traffic"] = traffic["roofline"]
        result["roofline"]["traffic_source"] = "profiles/r01_b_pmc_traffic.json (raw FETCH_SIZE+WRITE_SIZE, bytes per launch)"
    result["roofline_gather"] = {"kernel": "leaf_gather_deepsea_kernel", "bound": "hbm", "achieved": gach, "peak": PEAK_HBM_GBPS,
                                 "unit": "GB/s", "frac": gach / PEAK_HBM_GBPS, "traffic": None, "avg_launch_us": 1e3 * g_ms,
                                 "launches": prof["gather"]["launches"], "bytes_per_launch": gbytes}
    if "roofline_gather" in traffic:
        result["roofline_gather"]["traffic"] = traffic["roofline_gather"]
    result["kernel_ms_instrumented_pass"] = {k: round(v["ms"], 3) for k, v in prof.items()}
    result["instrumented_pass_env_steps_per_s"] = B * T * args.steps / dt_instr

    # ---- CPU baseline: the oracle on a bounded sample of the same workload, rank 0 at N = 1 only -------------------
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import eaz_oracle as O

        cb, ct = args.cpu_sample_batch, args.cpu_sample_steps
        ncores = cpu_share()
        O.lib().eazo_set_num_threads(ncores)
        oenv = O.OracleEnv.deepsea(G, am)
        onet = O.OracleNet(G * G, H, A, eaz.params.fc_weight_list(params), bitset=hstate[eaz.params.HASH_MODULE]["binary_set"])
        ost = oenv.init(cb)
        c0 = time.perf_counter()
        O.selfplay(oenv, onet, ost, T=ct, S=S, key=(0, 0))
        cdt = time.perf_counter() - c0
        result["cpu_baseline"] = {"value": cb * ct / cdt, "unit": "env-steps/s", "cores": O.lib().eazo_num_threads(),
                                  "kind": "port", "sample": f"DeepSea-{G}, batch {cb}, {S} sims, {ct} real steps ({cdt:.1f} s)"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
