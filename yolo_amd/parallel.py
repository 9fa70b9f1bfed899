"""Batch sharding across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in CPU tests).  Mirrors split_render_data (yolo_modules/yolo_gluon.py:100-124) for the
one-process-per-device model: rank i owns batch[int(i*B/n):int((i+1)*B/n)].

Inference is embarrassingly parallel: no collective on the data path, results are gathered on
rank 0 only when the caller asks.  The training exchange step is one SUM all-reduce of the flat
gradient bucket followed by the 1/global_batch rescale of trainer.step(batch_size) (car/YOLO.py:396).
"""
import torch
import torch.distributed as dist


def shard_bounds(batch_size, rank, world):
    return int(rank * batch_size / world), int((rank + 1) * batch_size / world)


def shard_batch(batch, rank=None, world=None):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    a, b = shard_bounds(len(batch), rank, world)
    return batch[a:b]


def gather_rows(rows, total, rank=None, world=None):
    """Gather per-image result rows (n_i, ...) of every rank on rank 0 in batch order."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    if world == 1:
        return rows
    bounds = [shard_bounds(total, r, world) for r in range(world)]
    bufs = [torch.empty((b - a,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device) for a, b in bounds]
    dist.all_gather(bufs, rows.contiguous()) if len({b - a for a, b in bounds}) == 1 else _uneven_gather(bufs, rows, rank, world)
    return torch.cat(bufs, dim=0) if rank == 0 else None


def _uneven_gather(bufs, rows, rank, world):
    for r in range(world):
        if r == rank:
            bufs[r].copy_(rows)
        dist.broadcast(bufs[r], src=r)


def max_over_ranks(value, device=None):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_sum_(flat):
    """In-place SUM all-reduce of a flat gradient bucket (no-op for a single rank)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat
