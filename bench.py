#!/usr/bin/env python3
"""bench.py -- self-play env-steps/sec of the HIP engine on BASELINE.json's headline configuration.

A "step" is one selfplay() call (src/selfplay.py:78-149): T real environment steps (config.py:37, T = 32) of a batch of B
environments per GPU with S Gumbel-MuZero simulations per real step.  env-steps = B_total * T per step (src/main.py:378);
search-internal steps are not counted.

--workload c2 (default, the configuration BASELINE.json's metric is quoted on): DeepSea-32, B = 4096, S = 64, FC H = 256.
--workload c3: Subleq word_size 16, B = 4096, S = 64, FC H = 256 (configs[2]).
--workload c4: connect_four, B = 8192, S = 200, ResNet 64ch x 5 blocks, two-player (configs[3]); T = 4 per step by default
               (one real step is 201 ResNet forwards of 8192 positions, ~0.6 s).

N > 1: `python bench.py --gpus N` starts the N ranks itself (one process per GPU, before anything touches the GPU); under torchrun the
process is a rank already.  The batch is sharded (weak scaling, B per GPU fixed); the only collective is one RCCL
all-gather per selfplay() call that pools the [T, B, ...] trajectory block of every rank (the counterpart of
src/main.py:383-385), inside the timed region.  When there are more ranks than visible GPUs (rehearsing the launch path on a
one-GPU box) the ranks share the card and the process group is gloo; the JSON line then carries "rehearsal".

Before anything is timed the parity gate runs 16 environments of the benchmark's inputs through the engine and the CPU oracle and
aborts on any difference.  Prints ONE JSON line on rank 0 (see the task contract), with `roofline` for the dominant kernel against the
f32-MFMA peak (C2: search_persist_kernel, C3: subleq16_selfplay_kernel, both one launch per selfplay() call; C4: resnet_forward_kernel),
measured live with HIP events on the launch stream, `leaf_gather` / `roofline_gather` for the leaf-gather class, `extra.c3` / `extra.c4`
(short passes of the other two configurations) plus `extra.c3_beta1` (directed exploration, beta = 1: the reference's experiment setting,
slurm_experiment_submitter.py:203,209) and `extra.c2_empty_set` (SURVEY 8d's second C2 input), each behind its own parity gate (default
run only), and `cpu_baseline` = the CPU oracle timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import socket  # noqa: E402
import subprocess  # noqa: E402

# numpy / torch / the engine are imported by run_rank() only: the self-launching parent of `--gpus N` must not touch the GPU
np = torch = eaz = engine = pool = None


def _imports():
    global np, torch, eaz, engine, pool
    import numpy as _np
    import torch as _torch

    import e_alphazero_amd as _eaz
    from e_alphazero_amd import engine as _engine, pool as _pool

    np, torch, eaz, engine, pool = _np, _torch, _eaz, _engine, _pool

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBPS = 8000.0        # MI355X_MICROARCH.md "HBM3E peak BW"
H = 256                       # config.py:24

WORKLOADS = {
    #        env-steps metric line (BASELINE.json)                 B     S    T   cpu sample (batch, real steps)
    "c2": dict(metric="self-play env-steps/sec (batch 4096, 64 sims)", B=4096, S=64, T=32, cpu=(4096, 8)),
    "c3": dict(metric="self-play env-steps/sec (Subleq-16, batch 4096, 64 sims)", B=4096, S=64, T=32, cpu=(1024, 4)),
    "c4": dict(metric="self-play env-steps/sec (connect_four, batch 8192, 200 sims, ResNet-5)", B=8192, S=200, T=4, cpu=(16, 1)),
}


def fc_flops_per_eval(in_dim, A, heads):
    f = 0
    for h in heads:
        out = 1 if h < 2 else A
        f += 2 * (in_dim * H + H * H + H * out)
    return f


def resnet_flops_per_eval(Hb=6, Wb=7, cin=2, C=64, nb=5, A=7):
    """2 * MACs of src/network/resnet.py:72-121 on one [6,7,2] board (all four heads)."""
    P = Hb * Wb
    macs = P * 9 * cin * C + 2 * nb * P * 9 * C * C                    # stem + tower
    macs += 2 * (P * C * 2 + 2 * P * A)                                # two policy heads: 1x1 conv(2) + linear(2P -> A)
    macs += 2 * (P * C * 1 + P * C + C)                                # value / ube heads: 1x1 conv(1) + linear(P -> C) + linear(C -> 1)
    return 2 * macs


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



def oracle_module():
    """The CPU oracle (checker only: parity gates and the cpu_baseline leg), its OpenMP team cut to the cores this process may use.  On a
    1-GPU box the process sees every hardware thread of the host but owns a 16-core quota: a default-sized team's idle workers spin through
    that quota after each parallel region and the cgroup then throttles the whole process -- the launching thread included (a 1.4 ms
    reanalyze() call measured 31 ms right behind a parity gate)."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # read by libgomp when it is loaded
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import eaz_oracle as O

    O.lib().eazo_set_num_threads(cpu_share())
    return O

def build_workload(name, args, device, rank, beta=0.0):
    """-> dict(env, net, search, selfplay kwargs, description strings, oracle factory for the CPU baseline).
    beta != 0: directed exploration (config.py:38,172): exploration policy head in the search, beta * linspace(0, 1, B) per environment."""
    w = dict(WORKLOADS[name])
    B = args.batch or w["B"]
    S = args.sims or w["S"]
    T = args.selfplay_steps or w["T"]
    explore = beta != 0.0
    sp_kw = dict(directed_exploration=True, exploration_beta=float(beta)) if explore else {}
    if name == "c2":
        G, A = args.grid, 2
        # synthetic inputs: action_map ~ Bernoulli(0.5) seed 0, random-init FC net seed 0, half-filled bit set
        am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
        params = eaz.params.init_fc_params(G * G, H, A, seed=0)
        hstate = eaz.params.init_hash_state(fill=args.hash_fill, seed=0)
        env = engine.EnvHandle.deepsea(G, am, device=device)
        net = engine.FCNet(env.ctx, G * G, H, A, params, hstate)
        search = engine.Search(env, net, B, S, discount=0.997, exploration=explore)
        desc = f"DeepSea size={G}, batch={B}/GPU, {S} sims, {T} real steps per selfplay() call, FC H={H} (BASELINE.json configs[1])"
        fill_txt = "empty" if args.hash_fill == 0 else ("half-filled" if args.hash_fill == 0.5 else f"{args.hash_fill:g}-filled")
        data = f"synthetic (seeded DeepSea action map, random-init FC net, {fill_txt} novelty set)"
        net_kernel, net_flops = "fc16_forward_kernel", fc_flops_per_eval(G * G, A, (0, 1, 2)) * B
        pmc_tag = f"DeepSea-{G}, B={B}, {S} sims"
        gather_kernel = "leaf_gather_deepsea_kernel"

        def oracle(O):
            oenv = O.OracleEnv.deepsea(G, am)
            onet = O.OracleNet(G * G, H, A, eaz.params.fc_weight_list(params), bitset=hstate[eaz.params.HASH_MODULE]["binary_set"])
            return oenv, onet, dict(discount=0.997), f"DeepSea-{G}"
    elif name == "c3":
        W = args.word_size
        cols, rows = (W - 1).bit_length() + 1, W + 32                   # binary encoding, envs/subleq.py:59-60,697-705
        in_dim, hash_off = rows * cols, W * cols                        # hash_io: rows W.. only (fully_connected.py:85-87)
        params = eaz.params.init_fc_params(in_dim, H, W, seed=0)
        hstate = eaz.params.init_hash_state(fill=args.hash_fill, seed=0)
        env = engine.EnvHandle.subleq(W, (1,), True, device=device)     # task NEGATION_POSITIVE (config.py:18)
        # layer-1 input order: a switch of the engine's numerics contract (include/eaz.h eaz_net_set_input_order) that the oracle follows
        # (parity gate, cpu_baseline).  "constants-first" (the default at word_size 16, where the one-launch kernel profits): the chain of
        # layer 1 sums example_input / example_output -- constants of an episode -- first; "ascending": the plain k-ascending chain.
        cfirst = args.l1_order == "constants-first" or (args.l1_order == "auto" and W == 16)
        order = eaz.params.subleq_constants_first(W, True) if cfirst else None
        net = engine.FCNet(env.ctx, in_dim, H, W, params, hstate, hash_off=hash_off, hash_len=in_dim - hash_off, input_order=order)
        search = engine.Search(env, net, B, S, discount=0.97, exploration=explore)
        A = W
        desc = (f"Subleq word_size={W} task=1 binary obs, batch={B}/GPU, {S} sims, {T} real steps per selfplay() call, FC H={H} "
                + ("(BASELINE.json configs[2])" if W == 16 else "(configs[2] at another word size: the per-simulation kernels)")
                + (", layer-1 input order: constants first (numerics switch eaz_net_set_input_order; the oracle follows it)" if cfirst else
                   ", layer-1 input order: ascending (default contract)"))
        w["l1_exec_inputs"] = (144 if cfirst else 224) if W == 16 else in_dim   # inputs the one-launch kernel's layer 1 chains per forward
        data = "synthetic (random-init FC net, half-filled novelty set)"
        # dense rows: narrow heads run the pipelined 16-row kernel, wide heads (W > 16) the 64-row tiles
        net_kernel, net_flops = "fc16p_forward_kernel" if W == 16 else "fc16_forward_kernel | fc_forward_kernel (by sub-batch size)", fc_flops_per_eval(in_dim, A, (0, 1, 2)) * B
        pmc_tag = f"Subleq-{W}, B={B}, {S} sims"
        gather_kernel = ("subleq16_tree_leaf_kernel (tree step + env step + observation row + novelty)" if W == 16 else
                         "subleq_tree_leaf_kernel (tree step + env step + observation row + novelty, one wavefront per tree)" if W >= 64 else
                         "subleq_wave_kernel (env step + observation row + novelty)")
        w["metric"] = w["metric"].replace("Subleq-16", f"Subleq-{W}")

        def oracle(O):
            oenv = O.OracleEnv.subleq(W, (1,), True)
            onet = O.OracleNet(in_dim, H, W, eaz.params.fc_weight_list(params), hash_off=hash_off, hash_len=in_dim - hash_off,
                               bitset=hstate[eaz.params.HASH_MODULE]["binary_set"])
            onet.set_input_order(order)
            return oenv, onet, dict(discount=0.97), f"Subleq-{W}"
    else:
        nb, A = 5, 7
        params, state = eaz.params.init_resnet_params(6, 7, 2, A, 64, nb, seed=0)
        env = engine.EnvHandle.connect4(device=device)
        net = engine.ResNet(env.ctx, 6, 7, 2, A, params, state, num_blocks=nb)
        search = engine.Search(env, net, B, S, discount=0.997, two_players_game=True)
        desc = f"connect_four, batch={B}/GPU, {S} sims, {T} real steps per selfplay() call, ResNet 64ch x {nb} blocks, two-player (BASELINE.json configs[3])"
        data = "synthetic (random-init ResNet with synthetic BatchNorm statistics, empty novelty set)"
        net_kernel, net_flops = "resnet_forward_kernel", resnet_flops_per_eval(nb=nb, A=A) * B
        pmc_tag, gather_kernel = None, "connect4_leaf_kernel (env.step on bitboards + child state + observation row + novelty lookup, one launch)"

        def oracle(O):
            oenv = O.OracleEnv.connect4()
            onet = O.OracleNet.resnet(6, 7, 2, A, eaz.params.resnet_array_list(params, state, nb), num_blocks=nb)
            return oenv, onet, dict(discount=0.997, two_player=True), "connect_four"
    disc = {"c2": 0.997, "c3": 0.97, "c4": 0.997}[name]
    if explore:
        desc += f", directed exploration beta={beta:g}"
    return dict(w, sp_kw=sp_kw, beta=float(beta), name=name, B=B, S=S, T=T, A=A, env=env, net=net, search=search, desc=desc, data=data, net_kernel=net_kernel,
                net_flops=net_flops, pmc_tag=pmc_tag, gather_kernel=gather_kernel, oracle=oracle, discount=disc, two_player=name == "c4",
                in_dim=env.obs_dim)


def dry_run(args, world, rank):
    """Everything of the N-rank path that needs no device: process group (gloo), one packed trajectory block per rank with the
    workload's record layout, the all-gather, the barrier / max-over-ranks timing and the JSON line.  Not a measurement."""
    import torch as _torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.join(ROOT, "e-alphazero_amd"))
    import pool as _pool   # the package __init__ loads libeaz.so; the pooling helpers alone need only torch

    w = WORKLOADS[args.workload]
    B, T = args.batch or w["B"], args.selfplay_steps or w["T"]
    state_bytes, A = {"c2": (16, 2), "c3": (336, 16), "c4": (80, 7)}[args.workload]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    def fake_output(r, step):
        """a SelfplayOutput-shaped dict of rank r at step `step` (different data every step: ordering / buffer-reuse bugs would show)"""
        g = _torch.Generator().manual_seed(1000 * step + r)
        f = lambda *sh: _torch.randn(*sh, generator=g)
        return dict(state=_torch.randint(0, 256, (T, B, state_bytes), dtype=_torch.uint8, generator=g), root_value=f(T, B),
                    root_epistemic_std=f(T, B), value_prediction=f(T, B), ube_prediction=f(T, B), q_values_epistemic_variance=f(T, B, A),
                    action=_torch.randint(0, A, (T, B), dtype=_torch.int32, generator=g))

    pooler = _pool.TrajectoryPooler() if world > 1 else None
    R = _pool.record_bytes(state_bytes, A)
    pend = []
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for step in range(args.steps):
        if world > 1:
            pend.append(pooler.submit(fake_output(rank, step)))   # the bench's one_step() without the device work
    if world > 1:
        pooler.flush()
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = True
    if world > 1:
        tmax = _torch.tensor([dt], dtype=_torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # the two most recent steps are still in their landing buffers: every rank's block must be where main.py:383-385 puts it
        for step in range(max(0, args.steps - 2), args.steps):
            pooled = _pool.pooled_batch_major(pend[step].wait())           # [world * B, T, R]
            ok = ok and tuple(pooled.shape) == (world * B, T, R)
            for r in range(world):
                ok = ok and bool(_torch.equal(pooled[r * B:(r + 1) * B], _pool.pack(fake_output(r, step)).permute(1, 0, 2)))
    if rank == 0:
        print(json.dumps({"metric": w["metric"], "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)",
                          "config": {"workload": f"dry run of {args.workload}", "global_batch": B * world,
                                     "parallelism": f"batch-sharded x{world}, gloo all-gather of trajectories" if world > 1 else "single process"},
                          "dry_run": "launch path + trajectory all-gather only, no device work: not a measurement",
                          "pooled_block_ok": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("dry run: pooled trajectory block mismatch")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--word-size", type=int, default=16, help="c3: Subleq word size (16 = configs[2]; the reference's code default is 256, main.py:159)")
    ap.add_argument("--l1-order", choices=("auto", "ascending", "constants-first"), default="auto",
                    help="c3: layer-1 input order of the FC network (numerics switch, include/eaz.h eaz_net_set_input_order); auto = constants-first at word_size 16")
    ap.add_argument("--grid", type=int, default=32, help="DeepSea size (c2)")
    ap.add_argument("--batch", type=int, default=0, help="environments per GPU (default: the workload's)")
    ap.add_argument("--sims", type=int, default=0)
    ap.add_argument("--selfplay-steps", type=int, default=0)
    ap.add_argument("--hash-fill", type=float, default=0.5, help="fraction of the novelty bit-set that is set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short C3 / C4 passes appended to the default C2 run")
    ap.add_argument("--cpu-sample-batch", type=int, default=0)
    ap.add_argument("--cpu-sample-steps", type=int, default=0)
    ap.add_argument("--dry-run", action="store_true",
                    help="no device work: argument handling, the self-launch of N ranks, process-group set-up, the trajectory "
                         "all-gather (gloo) and the JSON line only; `value` is null (used by the CPU test of the N > 1 path)")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (one per GPU) with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and wait for them.  Runs before this process has imported torch
    or touched the GPU; the ranks are children (never an exec of a process that initialised HIP).  Rank 0 prints the JSON line."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks' processes on this stack
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:   # a rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args.gpus, argv))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']} (launch with "
                         f"--nproc-per-node {args.gpus}, or without torchrun: bench.py starts the ranks itself)")
    run_rank(args)


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before HIP initialises (torchrun ranks inherit the caller's environment)
    if args.dry_run:
        return dry_run(args, world, rank)
    _imports()
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a ROCm device: the engine has no CPU path")
    rehearsal = world > 1 and ndev < world  # more ranks than cards: the ranks share GPUs and talk over gloo
    device = local_rank % ndev
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    n_gpus = world
    dev = torch.device("cuda", device)

    if dist is not None:
        # communicator set-up (RCCL builds its rings lazily inside the first collective) is not part of any step
        warm = torch.zeros(1, device="cpu" if rehearsal else dev)
        dist.all_reduce(warm)
        if not rehearsal:
            torch.cuda.synchronize()

    wl = build_workload(args.workload, args, device, rank)
    B, S, T = wl["B"], wl["S"], wl["T"]

    # ---- parity gate ("parity asserted before timing"): the engine against the CPU oracle on a 16-environment subset
    # of the benchmark's inputs (same env / weights / novelty set, the first 16 initial states), rank 0 at N = 1 ----
    parity = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        parity = parity_gate(wl, args)

    main_oracle, main_cpu = wl["oracle"], wl["cpu"]
    meas = measure(wl, args, rank, dist, rehearsal, dev, steps=args.steps, warmup=args.warmup)
    dt = meas["dt"]
    env_steps = B * n_gpus * T * args.steps
    value = env_steps / dt if dt > 0 else 0.0
    result = {
        "metric": wl["metric"],
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
        "data": wl["data"],
        "config": {"workload": wl["desc"], "global_batch": B * n_gpus,
                   "parallelism": f"batch-sharded x{n_gpus}, RCCL all-gather of trajectories" if n_gpus > 1 else "single GPU"},
    }
    if rehearsal:
        result["rehearsal"] = f"{world} ranks share {ndev} GPU(s), gloo all-gather through host memory: launch-path check, not a scaling number"
    result.update(meas["fields"])
    if parity is not None:
        result["parity_gate"] = parity

    # ---- the other single-GPU configurations of BASELINE.json, short passes after the timed region (driver-visible C3 / C4 numbers) ----
    if rank == 0 and n_gpus == 1 and args.workload == "c2" and not args.no_extra and not args.batch and not args.sims:
        result["extra"] = {}
        #                tag              workload (steps, warmup, T)  beta  hash fill
        for tag, name, (xs, xw, xt), xbeta, xfill in (("c3", "c3", (3, 1, 0), 0.0, None), ("c3_beta1", "c3", (3, 1, 0), 1.0, None),
                                                      ("c3_ascending", "c3", (3, 1, 0), 0.0, None),
                                                      ("c2_empty_set", "c2", (3, 1, 0), 0.0, 0.0), ("c4", "c4", (1, 1, 8), 0.0, None)):
            xargs = argparse.Namespace(**vars(args))
            xargs.workload, xargs.selfplay_steps = name, xt
            if tag == "c3_ascending":   # the default numerics contract (k-ascending layer-1 chain): extra.c3 runs the constants-first switch
                xargs.l1_order = "ascending"
            if xfill is not None:
                xargs.hash_fill = xfill
            wl = meas = None
            torch.cuda.empty_cache()
            try:  # a side measurement must never cost the headline line
                wl = build_workload(name, xargs, device, rank, beta=xbeta)
                gate = parity_gate(wl, xargs) if not args.no_cpu_baseline else None   # raises: the pass is then reported as an error, not timed
                meas = measure(wl, xargs, rank, None, False, dev, steps=xs, warmup=xw)
                result["extra"][tag] = dict({"metric": wl["metric"], "value": wl["B"] * wl["T"] * xs / meas["dt"], "unit": "env-steps/s",
                                             "steps": xs, "warmup": xw, "ms_per_step": 1e3 * meas["dt"] / xs,
                                             "config": {"workload": wl["desc"]}, "data": wl["data"], "parity_gate": gate}, **meas["fields"])
            except (Exception, SystemExit) as exc:  # noqa: BLE001
                result["extra"][tag] = {"error": f"{type(exc).__name__}: {exc}"}
        wl = meas = None
        # ---- reanalyze() and evaluate() on C2's and C3's inputs (own parity gates; never at the headline's cost) ----
        for tag, name in (("next_c2", "c2"), ("next_c3", "c3")):
            if args.no_cpu_baseline:
                break
            xargs = argparse.Namespace(**vars(args))
            xargs.workload, xargs.selfplay_steps = name, 0
            torch.cuda.empty_cache()
            try:
                wl = build_workload(name, xargs, device, rank)
                wl["search"] = None
                result["extra"][tag] = next_rows(wl, xargs)
            except (Exception, SystemExit) as exc:  # noqa: BLE001
                result["extra"][tag] = {"error": f"{type(exc).__name__}: {exc}"}
            wl = None

    # ---- CPU baseline: the oracle on a bounded sample of the same workload, rank 0 at N = 1 only -------------------
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        O = oracle_module()
        cb = args.cpu_sample_batch or main_cpu[0]
        ct = args.cpu_sample_steps or main_cpu[1]
        ncores = cpu_share()
        O.lib().eazo_set_num_threads(ncores)
        oenv, onet, okw, oname = main_oracle(O)
        ost = oenv.init(cb)
        c0 = time.perf_counter()
        O.selfplay(oenv, onet, ost, T=ct, S=S, key=(0, 0), **okw)
        cdt = time.perf_counter() - c0
        result["cpu_baseline"] = {"value": cb * ct / cdt, "unit": "env-steps/s", "cores": O.lib().eazo_num_threads(),
                                  "kind": "port", "sample": f"{oname}, batch {cb}, {S} sims, {ct} real steps ({cdt:.1f} s)"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def parity_gate(wl, args):
    """16 environments of the benchmark's inputs through the engine and through the CPU oracle: actions and states must be bit-identical,
    values within 1e-5 (the north-star tolerance).  Raises before anything is timed if they are not."""
    O = oracle_module()
    nb, nt = 16, 2 if wl["name"] != "c4" else 1
    S = wl["S"] if wl["name"] != "c4" else 32      # connect_four on the CPU: a shorter search keeps the gate at a few seconds
    kw = dict(discount=wl["discount"], two_players_game=wl["two_player"], exploration=wl["beta"] != 0.0)
    search = engine.Search(wl["env"], wl["net"], nb, S, **kw)
    st = wl["env"].init(nb, key=0)
    out = search.selfplay(st, nt, key=4242, **wl["sp_kw"])
    torch.cuda.synchronize()
    oenv, onet, okw, oname = wl["oracle"](O)
    ost = oenv.init(nb)
    ref = O.selfplay(oenv, onet, ost, T=nt, S=S, key=engine.key_pair(4242), **okw, **wl["sp_kw"])
    # Subleq: the oracle keeps words as int32, the device blob of include/eaz.h as bytes -- same fields, compared in the oracle's layout
    conv = (lambda a: O.subleq_device_to_oracle(a, wl["A"])) if wl["name"] == "c3" else (lambda a: a)
    ok_int = bool(np.array_equal(out["action"].cpu().numpy(), ref["action"]) and np.array_equal(conv(out["state"].cpu().numpy()), ref["state"])
                  and np.array_equal(conv(st.cpu().numpy()), ost))
    err = max(float(np.max(np.abs(out[k].cpu().numpy() - ref[k]))) for k in
              ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction", "q_values_epistemic_variance"))
    if not ok_int or not err <= 1e-5:
        raise SystemExit(f"bench.py parity gate FAILED on {oname}: integer results identical = {ok_int}, max float error = {err:.3g}")
    del search
    if wl["beta"]:
        oname += f", directed exploration beta={wl['beta']:g}"
    return {"checked": f"{oname}: {nb} environments x {nt} real steps x {S} simulations, engine vs CPU oracle (in-repo restatement; emctx parity unpinned)",
            "actions_states_bit_identical": ok_int, "max_abs_float_error": err, "tolerance": 1e-5}


def next_rows(wl, args, calls=3):
    """SURVEY 8(f) N1 / N2 on the workload's inputs, each parity-gated on 16 environments first: `reanalyze()` (src/reanalyze.py:52-131: one
    search per sampled state with beta = reanalyze_beta, then the target post-processing) as searched states per second, `evaluate()`
    (src/evaluate.py:13-57: greedy episodes until every environment has terminated) as real env-steps per second.  -> dict"""
    O = oracle_module()
    env, net, B, S, A = wl["env"], wl["net"], wl["B"], wl["S"], wl["A"]
    kw = dict(discount=wl["discount"], two_players_game=wl["two_player"])
    oenv, onet, okw, oname = wl["oracle"](O)
    conv = (lambda a: O.subleq_device_to_oracle(a, A)) if wl["name"] == "c3" else (lambda a: a)
    back = (lambda a: O.subleq_oracle_to_device(a)) if wl["name"] == "c3" else (lambda a: a)
    out = {}

    # ---- N1 reanalyze: mid-episode states (5 random real steps from init) and their successors under a random action --------------------
    def sample(n):
        ost, _ = O.random_play(oenv, n, 5, key=(11, 0))
        act = np.random.default_rng(3).integers(0, A, n).astype(np.int32)
        onxt = oenv.step(ost.copy(), act, key=(12, 0), autoreset=False)
        return ost, onxt
    nb = 16
    ost, onxt = sample(nb)
    g = O.gumbel((21, 0), nb * A).reshape(nb, A)
    rs = engine.Search(env, net, nb, S, **kw)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    got = rs.reanalyze(cu(back(ost)), cu(back(onxt)), key=5, gumbel=cu(g), reanalyze_beta=0.5)
    torch.cuda.synchronize()
    ref = O.reanalyze(oenv, onet, ost, onxt, S, g, reanalyze_beta=0.5, **okw)
    ok_int = bool(np.array_equal(got["action"].cpu().numpy(), ref["action"]))
    err = max(float(np.max(np.abs(got[k].cpu().numpy() - ref[k]))) for k in
              ("value_target", "ube_target", "exploration_policy_target", "exploitation_policy_target"))
    if not ok_int or not err <= 1e-5:
        raise SystemExit(f"reanalyze parity gate FAILED on {oname}: actions identical = {ok_int}, max float error = {err:.3g}")
    gate = {"checked": f"{oname}: {nb} sampled states x {S} simulations, reanalyze() engine vs CPU oracle", "actions_bit_identical": ok_int,
            "max_abs_float_error": err, "tolerance": 1e-5}
    del rs
    ost, onxt = sample(B)
    st, nx = cu(back(ost)), cu(back(onxt))
    rs = engine.Search(env, net, B, S, **kw)
    rs.reanalyze(st, nx, key=0, reanalyze_beta=0.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(calls):
        rs.reanalyze(st, nx, key=1 + i, reanalyze_beta=0.5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["reanalyze"] = {"metric": f"reanalyzed states/s ({oname}, batch={B}, {S} sims, reanalyze_beta=0.5)", "value": B * calls / dt, "unit": "states/s",
                        "calls": calls, "ms_per_call": 1e3 * dt / calls, "launch_structure": rs.launch_info(), "parity_gate": gate,
                        "reference": "src/reanalyze.py:52-131 (SURVEY 8f N1)"}
    del rs

    # ---- N2 evaluate: greedy episodes from init; the gate stops after four iterations (a whole episode on the CPU takes too long here) -----
    es = engine.Search(env, net, nb, S, gumbel_scale=0.0, **kw)
    st16, ost16 = env.init(nb, key=0), oenv.init(nb)
    ret, it = es.evaluate(st16, exploitation_beta=0.0, max_episode_length=3)
    oret, oit = O.evaluate(oenv, onet, ost16, S=S, exploitation_beta=0.0, max_episode_length=3, **okw)
    ok_int = bool(it == oit and np.array_equal(conv(st16.cpu().numpy()), ost16))
    err = float(np.max(np.abs(ret.cpu().numpy() - oret)))
    if not ok_int or not err <= 1e-5:
        raise SystemExit(f"evaluate parity gate FAILED on {oname}: states / trip count identical = {ok_int}, max float error = {err:.3g}")
    gate = {"checked": f"{oname}: {nb} environments x {it} iterations x {S} simulations, evaluate() engine vs CPU oracle", "states_bit_identical": ok_int,
            "max_abs_float_error": err, "tolerance": 1e-5}
    del es
    es = engine.Search(env, net, B, S, gumbel_scale=0.0, **kw)
    es.evaluate(env.init(B, key=0), exploitation_beta=0.0, max_episode_length=1000)
    torch.cuda.synchronize()
    steps = 0
    t0 = time.perf_counter()
    for i in range(calls):
        _, it = es.evaluate(env.init(B, key=0), exploitation_beta=0.0, max_episode_length=1000)
        steps += it
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["evaluate"] = {"metric": f"evaluate env-steps/s ({oname}, batch={B}, {S} sims, greedy search, whole episodes)", "value": B * steps / dt,
                       "unit": "env-steps/s", "calls": calls, "iterations_per_call": steps / calls, "ms_per_call": 1e3 * dt / calls,
                       "launch_structure": es.launch_info(), "parity_gate": gate, "reference": "src/evaluate.py:13-57 (SURVEY 8f N2)"}
    return out


def measure(wl, args, rank, dist, rehearsal, dev, steps, warmup):
    """warmup + EXACTLY `steps` timed selfplay() calls (barrier + synchronize on both sides, max over ranks), then the instrumented pass
    that feeds `roofline`.  -> dict(dt, fields)"""
    env, search, B, S, T, name = wl["env"], wl["search"], wl["B"], wl["S"], wl["T"], wl["name"]
    states = env.init(B, key=rank)
    outputs = None
    pooler = pool.TrajectoryPooler(to_host=rehearsal) if dist is not None else None

    def one_step(i):
        nonlocal outputs
        outputs = search.selfplay(states, T, key=(1000 * rank + i), outputs=outputs, **wl["sp_kw"])
        if pooler is not None:
            # pool the trajectories of all ranks (replay-buffer landing zone): ONE all-gather of the packed block per call, started
            # asynchronously on a double-buffered block so that it overlaps the NEXT selfplay() launch; the timed region
            # ends with every collective complete (sync() below)
            pooler.submit(outputs)

    def sync():
        if pooler is not None:
            pooler.flush()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        one_step(-1 - i)
    if warmup == 0:
        # graph capture, kernel module loading and (N > 1) the all-gather's buffers are one-time set-up: with --warmup 0 run one
        # single real step through the same path before the clock starts (not a step of the workload: T = 1)
        search.selfplay(states, 1, key=(1000 * rank + 999), **wl["sp_kw"])
        if pooler is not None:
            pool.TrajectoryPooler(to_host=rehearsal).submit(search.selfplay(states, 1, key=(1000 * rank + 998), **wl["sp_kw"])).wait()
    # ---- timed region: EXACTLY K steps in the production configuration (one captured hipGraph per real step) ----
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        one_step(i)
    ev1.record()   # this rank's own launch stream is done here (the collectives of the last two steps may still be in flight)
    sync()
    dt = time.perf_counter() - t0
    rank_ms = None
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # per-rank time of the K steps on the rank's own launch stream (HIP events): the first real scaling run shows a straggling GPU
        # without a second run -- value / ms_per_step stay the barrier-to-barrier maximum over the ranks
        own = torch.tensor([ev0.elapsed_time(ev1) / max(steps, 1)], dtype=torch.float64, device="cpu" if rehearsal else dev)
        allr = [torch.zeros_like(own) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, own)
        rank_ms = [float(x.item()) for x in allr]
    # ---- instrumented pass: the same kernels again with HIP events recorded on the launch stream around every kernel
    # (event records cannot sit between the nodes of a replayed graph, so this pass uses plain stream launches; kernel
    # durations are unaffected, only the host-side launch gaps differ).  Feeds `roofline`, never `value`. ----
    search.profile(True)
    search.profile_read(reset=True)
    sync()
    t1 = time.perf_counter()
    n_instr = min(steps, 2) if name != "c4" else 1
    T_instr = T if name != "c4" else 1
    for i in range(n_instr):
        search.selfplay(states, T_instr, key=(1000 * rank + 500 + i), outputs=outputs if T_instr == T else None, use_graph=False, **wl["sp_kw"])
    sync()
    dt_instr = time.perf_counter() - t1
    prof = search.profile_read(reset=True)
    search.profile(False)
    persistent = prof["search"]["launches"] > 0
    fields = {}
    # ---- roofline of the dominant kernel.  FLOPs counted are the ones the kernel EXECUTES (the three heads self-play reads: value, ube,
    # one policy head): layers 2-3 dense; layer 1 of a one-hot DeepSea row is a row gather (0 FLOPs, exact: every other term of the
    # chain is fma(0, w, acc) = acc) and on dense rows only the non-zero 64x4 blocks.  `dense_equiv` prices the same launches at the
    # dense count of SURVEY.md 8(d) (2 * MACs of every layer).  ResNet: dense implicit GEMMs, executed == dense. ----
    heads = (0, 1, 2)
    A = wl["A"]
    l23 = sum(2 * (H * H + H * (1 if h < 2 else A)) for h in heads) * B
    dense_flops = wl["net_flops"]
    if persistent and name == "c3" and prof["search"]["launches"] == n_instr:
        # subleq16_selfplay_kernel: ONE launch per selfplay() call = T real steps x (root inference + S simulations): tree phase on LDS node
        # records, all-scalar interpreter, forward of 3 heads per simulation (16 trees per workgroup).  EXECUTED FLOPs: the (S + 1) forwards per
        # real step with layer 1 over the inputs the kernel chains -- 224 of the 240 (the pad flags of the memory words are structurally zero and
        # dropped, exact) or, under the constants-first input order, 144 (the 80 example-row inputs enter as one precomputed partial sum per
        # tree); the zero-padded k-steps of the last half-round are not counted.  `dense` prices the same forwards at all 240 inputs.
        kname = "subleq16_selfplay_kernel"
        n_launch = prof["search"]["launches"]
        k_ms = search.profile_replay("search", 3)
        pair_us = 1e3 * prof["search"]["ms"] / n_launch
        k1 = wl.get("l1_exec_inputs", wl["in_dim"])
        exec_row = sum(2 * (k1 * H + H * H + H * (1 if h < 2 else A)) for h in heads)
        flops, dense = T_instr * (S + 1) * exec_row * B, T_instr * (S + 1) * dense_flops
        algo = (f"{T_instr} real steps x ({S} simulations + root) x {B} rows x forward of 3 heads as executed ({exec_row} FLOP per row: layer 1 over "
                f"{k1} of the {wl['in_dim']} inputs; dense: {dense_flops // B}); the launch also walks the trees and runs the Subleq interpreter, "
                "during which the MFMA pipes idle")
    elif persistent and name == "c3":
        # subleq16_search_kernel (EAZ_TUNE_PERSIST = 2): all S simulations of a real step in one launch (tree step + interpreter + forward per simulation, 16 trees per
        # workgroup).  A ~3 ms launch: the event pair of the instrumented pass is its duration (the ~4 us launch gap is 0.1 %).  FLOPs: the S
        # forwards, dense (the kernel runs whole rounds of layer 1; padded k-steps are not counted).
        kname = "subleq16_search_kernel"
        n_launch = prof["search"]["launches"]
        k_ms = pair_ms = prof["search"]["ms"] / n_launch
        flops = dense = S * dense_flops
        pair_us = 1e3 * pair_ms
        algo = (f"{S} simulations x {B} leaves x dense forward of 3 heads ({dense_flops // B} FLOP per leaf); the launch also walks the trees and "
                "runs the Subleq interpreter, during which the MFMA pipes idle")
    elif persistent:
        # search_persist_kernel: ALL S simulations of a real step in one launch (trees in LDS, DeepSea step, S forwards of B leaves)
        # self-play: ONE launch per selfplay() call = T real steps x (root inference + S simulations), trees in LDS, DeepSea steps,
        # (S + 1) forwards of B rows per real step
        kname = "search_persist_kernel"
        n_launch = prof["search"]["launches"]
        fused = n_launch == n_instr
        k_ms = search.profile_replay("search", 3 if fused else 20)
        n_fwd = T_instr * (S + 1) if fused else S
        flops, dense = n_fwd * l23, n_fwd * dense_flops
        pair_us = 1e3 * prof["search"]["ms"] / n_launch
        algo = (f"{T_instr} real steps x ({S} simulations + root) x {B} rows x (layers 2-3 of 3 heads = {l23 // B} FLOP per row; "
                "layer 1 is a one-hot row gather)") if fused else \
               f"{S} simulations x {B} leaves x (layers 2-3 of 3 heads = {l23 // B} FLOP per leaf; layer 1 is a one-hot row gather)"
    else:
        kname = wl["net_kernel"]
        k_ms = search.profile_replay("net", 200 if name != "c4" else 10)
        n_launch = max(prof["net"]["launches"], 1)
        if kname.startswith("fc"):
            # counter unit: one k-step of a 16-row tile; never more than the dense layer 1 (padded k-steps of a partial block are not work)
            l1 = min(prof["net"]["layer1_ksteps"] * (16 * H * 4 * 2) / n_launch, sum(2 * wl["in_dim"] * H for _ in heads) * B)
            flops = l23 + l1
        else:
            flops = dense_flops
        dense = dense_flops
        pair_us = 1e3 * prof["net"]["ms"] / n_launch
        algo = f"one forward of {B} rows"
    ach = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    roof = {"kernel": kname, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None, "avg_launch_us": 1e3 * k_ms,
            "avg_launch_us_event_pair_per_launch": pair_us, "launches": n_launch, "flops_per_launch": flops, "flops_counted": algo,
            "dense_flops_per_launch": dense, "dense_equiv_tflops": dense / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0}
    # HBM-side traffic per launch from the PMC passes committed under profiles/ (rocprofv3 FETCH_SIZE / WRITE_SIZE cannot be collected
    # from inside this process); only attached when the file was recorded for this workload and this kernel
    for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            continue
        k = next((v for kn, v in pmc.get("kernels", {}).items() if kname in kn), None)
        if wl["pmc_tag"] and pmc.get("workload") == wl["pmc_tag"] and k:
            # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies a wide coalesced streaming read (16 B per lane) at half its
            # bytes -> doubled (the kernel's fetches are its 16-B-per-lane weight streams); WRITE_SIZE is exact for 16-B-per-lane stores
            roof["traffic"] = 1024.0 * (2.0 * k.get("FETCH_SIZE_KB_per_launch", 0.0) + k.get("WRITE_SIZE_KB_per_launch", 0.0))
            roof["traffic_source"] = (f"profiles/{cand} (2 x FETCH_SIZE [gfx950 16-B/lane correction] + WRITE_SIZE, bytes per launch; recorded by "
                                      "rocprofv3 --pmc passes over this workload, not inside this process)")
            # tie the attached counters to the code that is being timed: the profile carries the source hash it was recorded on
            # (tools/pmc_summary.py), the library the hash it was built from (eaz_build_hash)
            from e_alphazero_amd import _lib
            rec, built = pmc.get("source_hash"), _lib.build_hash()
            roof["traffic_source_hash"] = rec
            roof["library_hash"] = built
            roof["traffic_source_status"] = "current" if rec == built else \
                (f"stale: recorded on sources {rec}, the library being timed was built from {built}" if rec else
                 f"stale: recorded before profiles carried a source hash (library: {built})")
            break
    fields["roofline"] = roof
    # ---- leaf gather (north-star: >= 40 % of the HBM roof on the leaf-gather kernel) ----
    gbytes = B * (2 * env.state_bytes + env.obs_dim)
    if prof["gather"]["launches"] == 0 and name == "c3":
        fields["leaf_gather"] = {
            "status": "retired for this configuration",
            "why": "Subleq-16: the search kernel keeps every node's state in LDS (32 bytes: the program as sixteen 4-bit words, input_after / "
                   "output_after, a header), steps the chosen edge with its all-scalar interpreter LDS to LDS, and hands the forward the observation "
                   "as 256 operand bytes per row in LDS -- no gather launch, no state or observation traffic to HBM in the timed path (DESIGN.md 4.4)"}
    elif prof["gather"]["launches"] == 0:
        fields["leaf_gather"] = {
            "status": "retired for this configuration",
            "why": "DeepSea's env.step is a few integer operations on a 16-byte state: the search kernel steps the chosen edge itself (child state "
                   "stays in LDS) and hands the forward the flat index of the one-hot observation -- no leaf-gather launch and no observation row "
                   "exists in the timed path.  At this batch size the standalone kernel of the unfused variant moves 4.3 MB per launch, below "
                   "launch latency (4.3 MB per launch: 0.06-0.07 of the HBM roof); with enough trees per launch to be bandwidth-bound the same kernel "
                   "reaches 0.43-0.44 of the 8 TB/s peak (>= 64 k trees, tools/gather_sweep.py, profiles/r04_d_gather_sweep.txt, DESIGN.md 5)"}
    else:
        sims = max(T_instr * n_instr * S, 1)
        g_ms = prof["gather"]["ms"] / sims
        gach = gbytes / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
        fields["roofline_gather"] = {"kernel": wl["gather_kernel"], "bound": "hbm", "achieved": gach, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                     "frac": gach / PEAK_HBM_GBPS, "traffic": None, "avg_us_per_simulation": 1e3 * g_ms,
                                     "bytes_per_simulation": gbytes,
                                     "note": "event pairs around the leaf-step launch(es) of one simulation (env.step + child state + observation row + "
                                             "novelty) against 2 x state + observation bytes per tree: a few hundred bytes per tree, i.e. 1-4 MB per "
                                             "launch -- far below what a launch needs to approach the HBM roof; the launch is bound by its latency "
                                             "(one dependent chain per environment: state load -> step -> stores), not by bandwidth (DESIGN.md 4.4, 5)"}
    if rank_ms is not None:
        fields["ms_per_step_by_rank"] = {"min": min(rank_ms), "max": max(rank_ms), "ranks": [round(x, 3) for x in rank_ms],
                                         "what": "HIP-event time of the K timed steps on each rank's own launch stream / K (no barrier inside)"}
    fields["launch_structure"] = search.launch_info()  # one_launch: the whole selfplay() call is ONE kernel launch (include/eaz.h eaz_search_launch_info)
    fields["kernel_ms_instrumented_pass"] = {k: round(v["ms"], 3) for k, v in prof.items()}
    fields["instrumented_pass_env_steps_per_s"] = B * T_instr * n_instr / dt_instr
    return dict(dt=dt, fields=fields)


if __name__ == "__main__":
    main()
