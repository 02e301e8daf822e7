"""Trajectory pooling across ranks: the counterpart of the reference's host-side concat of per-device self-play output
(src/main.py:383-385) before `buffer_fn.add`.

Self-play shards embarrassingly (one process per GPU, no collective inside the search); after each selfplay() call
every rank contributes its [T, B_local, R] block and receives every rank's block, [world, T, B_local, R], with ONE
all-gather (RCCL over xGMI on MI355X, gloo in the CPU tests).  All SelfplayOutput fields are packed into a single
uint8 record per (t, b) so that it is one collective, not six.

Landing layout: the reference's pmap output is [devices, T, B_local, ...] and `jnp.concatenate(jnp.swapaxes(x, 1, 2))`
(src/main.py:383-385) hands flashbax the batch-major sequence block [devices * B_local, T, ...], device-major along the batch
axis.  `pooled_batch_major` is that view of the all-gather output; `pooled_view` is the time-major [T, world * B_local, R] one.
"""
from __future__ import annotations

import torch

FLOAT_FIELDS = ("root_value", "root_epistemic_std", "value_prediction", "ube_prediction")


def record_bytes(state_bytes: int, num_actions: int) -> int:
    return state_bytes + 4 * len(FLOAT_FIELDS) + 4 * num_actions + 4


def pack(out: dict) -> torch.Tensor:
    """SelfplayOutput dict ([T,B,...] tensors) -> uint8 [T, B, record_bytes]."""
    T, B = out["root_value"].shape
    parts = [out["state"].reshape(T, B, -1)]
    parts += [out[k].contiguous().view(torch.uint8).reshape(T, B, 4) for k in FLOAT_FIELDS]
    parts.append(out["q_values_epistemic_variance"].contiguous().view(torch.uint8).reshape(T, B, -1))
    parts.append(out["action"].contiguous().view(torch.uint8).reshape(T, B, 4))
    return torch.cat(parts, dim=2).contiguous()


def unpack(block: torch.Tensor, state_bytes: int, num_actions: int) -> dict:
    """uint8 [T, B, record_bytes] -> SelfplayOutput dict."""
    T, B, R = block.shape
    assert R == record_bytes(state_bytes, num_actions)
    out, o = {}, 0
    out["state"] = block[:, :, o:o + state_bytes].contiguous()
    o += state_bytes
    for k in FLOAT_FIELDS:
        out[k] = block[:, :, o:o + 4].contiguous().view(torch.float32).reshape(T, B)
        o += 4
    out["q_values_epistemic_variance"] = block[:, :, o:o + 4 * num_actions].contiguous().view(torch.float32).reshape(T, B, num_actions)
    o += 4 * num_actions
    out["action"] = block[:, :, o:o + 4].contiguous().view(torch.int32).reshape(T, B)
    return out


def all_gather_trajectories(block: torch.Tensor, out: torch.Tensor | None = None, group=None) -> torch.Tensor:
    """block: uint8 [T, B, R] of this rank -> uint8 [world, T, B, R]: the blocks of all ranks, rank-major (the layout of the
    reference's pmap output, [devices, T, B_local, ...]).  `out` (same shape) is reused when given."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    T, B, R = block.shape
    if out is None:
        out = torch.empty(world, T, B, R, dtype=torch.uint8, device=block.device)
    assert tuple(out.shape) == (world, T, B, R) and out.dtype == torch.uint8 and out.is_contiguous(), "all_gather_trajectories: `out` must be a contiguous uint8 [world, T, B, R]"
    # concatenation-along-dim-0 form: accepted by both the RCCL and the gloo backend
    dist.all_gather_into_tensor(out.view(world * T, B, R), block, group=group)
    return out


class TrajectoryPooler:
    """The all-gather of step k off the critical path of step k + 1 ("off the critical path (overlap with next call)").

    `submit(out)` packs the SelfplayOutput of the call that has just been enqueued into one of TWO packed blocks and starts the all-gather
    with async_op=True: on RCCL the collective runs on the communicator's own stream behind the work already enqueued on the current
    stream (the pack), so the next selfplay() launch does not wait for it; the block / landing buffer of a slot is reused only after its
    previous collective has been waited for.  `Pending.wait()` makes the current stream (gloo: the host) wait for that step's collective
    and returns its [world, T, B, R] landing buffer, valid until two more submits.  `flush()` waits for everything in flight."""

    class Pending:
        def __init__(self, work, gathered, pooler=None, index=0):
            self.work, self.gathered, self.pooler, self.index = work, gathered, pooler, index

        def wait(self) -> torch.Tensor:
            # the landing buffer belongs to submit number `index`; the submit two steps later reuses it (double buffering)
            if self.pooler is not None and self.pooler.n - self.index > 2:
                raise RuntimeError(f"TrajectoryPooler: the landing buffer of submit {self.index} was overwritten by submit {self.index + 2} "
                                   f"({self.pooler.n} submits so far): wait() within two submits, or clone the result")
            if self.work is not None:
                self.work.wait()
                self.work = None
            return self.gathered

    def __init__(self, group=None, to_host: bool = False):
        self.group, self.to_host = group, to_host
        self.blocks = [None, None]
        self.landing = [None, None]
        self.pending = [None, None]
        self.n = 0

    def submit(self, out: dict) -> "TrajectoryPooler.Pending":
        import torch.distributed as dist

        slot = self.n & 1
        index = self.n
        if self.pending[slot] is not None:
            self.pending[slot].wait()  # the collective that read this slot's block two steps ago
        self.n += 1
        block = pack(out)
        if self.to_host:               # rehearsal on fewer GPUs than ranks: gloo through host memory
            block = block.cpu()
        if self.blocks[slot] is None or self.blocks[slot].shape != block.shape:
            self.blocks[slot] = torch.empty_like(block)
            world = dist.get_world_size(self.group)
            self.landing[slot] = torch.empty((world,) + tuple(block.shape), dtype=torch.uint8, device=block.device)
        self.blocks[slot].copy_(block)
        T, B, R = block.shape
        work = dist.all_gather_into_tensor(self.landing[slot].view(-1, B, R), self.blocks[slot], group=self.group, async_op=True)
        self.pending[slot] = TrajectoryPooler.Pending(work, self.landing[slot], self, index)
        return self.pending[slot]

    def flush(self) -> None:
        for p in self.pending:
            if p is not None:
                p.wait()


def pooled_view(gathered: torch.Tensor) -> torch.Tensor:
    """[world, T, B, R] -> time-major [T, world*B, R]"""
    W, T, B, R = gathered.shape
    return gathered.permute(1, 0, 2, 3).reshape(T, W * B, R)


def pooled_batch_major(gathered: torch.Tensor) -> torch.Tensor:
    """[world, T, B, R] -> [world*B, T, R]: `jnp.concatenate(jnp.swapaxes(x, 1, 2))` of src/main.py:383-385, the sequence
    layout the replay buffer is fed with (rank-major along the batch axis)."""
    W, T, B, R = gathered.shape
    return gathered.permute(0, 2, 1, 3).reshape(W * B, T, R)


def unpack_batch_major(block: torch.Tensor, state_bytes: int, num_actions: int) -> dict:
    """uint8 [B_total, T, R] -> SelfplayOutput dict of [B_total, T, ...] tensors."""
    out = unpack(block.permute(1, 0, 2), state_bytes, num_actions)
    return {k: v.transpose(0, 1).contiguous() for k, v in out.items()}
