"""Single-rank check of the RCCL calls bench.py makes at N > 1 (init with device_id, all_gather_into_tensor on uint8, all_reduce MAX,
barrier): run on the GPU box, `python tools/rccl_smoke.py`."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(12, dtype=torch.uint8, device="cuda").reshape(2, 3, 2)
out = torch.empty(1, 2, 3, 2, dtype=torch.uint8, device="cuda")
dist.all_gather_into_tensor(out.view(2, 3, 2), x)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
print("rccl single-rank ok", bool((out[0] == x).all()), float(t))
dist.destroy_process_group()
