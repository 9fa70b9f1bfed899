"""Batch sharding across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in CPU tests).  Mirrors split_render_data (yolo_modules/yolo_gluon.py:100-124) for the
one-process-per-device model: rank i owns batch[int(i*B/n):int((i+1)*B/n)].

Inference is embarrassingly parallel: no collective on the data path, results are gathered on
rank 0 only when the caller asks.  The training exchange step is one SUM all-reduce of the flat
gradient bucket followed by the 1/global_batch rescale of trainer.step(batch_size) (car/YOLO.py:396).
"""
import os

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


def global_batch_size(local_batch, device=None):
    """Sum of the ranks' local batch sizes: the `batch_size` of trainer.step(batch_size) (car/YOLO.py:396).  Ranks may
    hold uneven shards (shard_bounds), so this is a SUM all-reduce, not local * world."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return int(local_batch)
    t = torch.tensor([int(local_batch)], dtype=torch.int64, device=device if dist.get_backend() == 'nccl' else None)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def share_tuning(obj, src=0):
    """COLLECTIVE: rank `src`'s measured kernel choices (CarNet / Trainer .tuning_state()) -> every rank, so that all ranks
    launch the SAME kernel instantiations (tune='measure' times variants per box: left alone, N ranks pick N plans, and a
    rank on a slower variant is the straggler every all-reduce waits for).  Call it after rank `src` has built its plan
    (CarNet.plan_signature / Trainer.tune) and before the others build theirs.  Returns the state in force."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return obj.tuning_state()
    box = [obj.tuning_state() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    if dist.get_rank() != src:
        obj.load_tuning_state(box[0])
    return box[0]


def same_on_all_ranks(value):
    """COLLECTIVE: True when every rank holds an equal `value` (any picklable object)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return True
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, value)
    return all(g == got[0] for g in got)


def checkpoint_params(params):
    """The parameter dict a checkpoint holds: `.running_mean` / `.running_var` averaged over the ranks (copies; the
    live statistics stay local to a GPU: no SyncBN) -- what gluon's Parameter._reduce() does with the per-device
    copies when collect_params().save() writes a file (car/YOLO.py:549).  Collective: every rank must call it."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return params
    world = dist.get_world_size()
    out = dict(params)
    for n in sorted(params):
        if n.endswith(('.running_mean', '.running_var')):
            t = params[n].detach().clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            out[n] = t / world
    return out


def allreduce_sum_(flat):
    """In-place SUM all-reduce of a flat gradient bucket (no-op for a single rank)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def bucket_ranges(offsets, sizes, nbuckets, total=None):
    """Contiguous [start, end) ranges of a flat buffer, cut at parameter boundaries into ~equal buckets.
    offsets/sizes: start and length of every parameter's slot in the flat buffer (ascending)."""
    total = offsets[-1] + sizes[-1] if total is None else total
    target = max(1, -(-total // max(1, nbuckets)))
    ranges, start = [], 0
    for i, (o, s) in enumerate(zip(offsets, sizes)):
        end = offsets[i + 1] if i + 1 < len(offsets) else total      # a slot owns its alignment padding
        if end - start >= target or i + 1 == len(offsets):
            ranges.append((start, end))
            start = end
    return ranges


class GradBuckets(object):
    """Bucketed gradient all-reduce overlapped with the backward pass (one exchange step =
    allReduce(sum) of the flat gradient bucket).  The flat buffer is cut into contiguous buckets at parameter
    boundaries; `done(names)` is called as the backward finishes parameters and launches the asynchronous SUM
    all-reduce of every bucket whose parameters are all final, on the process group's own stream, while the
    remaining backward kernels keep running; `wait()` joins them before the optimiser.  xGMI is point-to-point,
    so a few large buckets (default 4 over 492 MB) keep every ring step bandwidth-bound.

    dtype='bf16' (last row): a bucket is cast to bf16 into a staging buffer, the staging buffer is
    SUM-reduced, and the sum is cast back into the fp32 gradients when the exchange is joined -- half the bytes per xGMI
    link (246 MB instead of 492), at the price of one bf16 rounding per rank's contribution and one per partial sum inside
    the collective -- a ring of N ranks rounds a partial sum at each of its N - 1 hops: |error| <= ~(N - 1) * 2^-8 * sum_r |g_r| per
    element (2^-8 * sum at N = 2, ~7x that at N = 8: small gradients added to a large partial sum lose their low bits); bounded
    against the fp32 exchange at world 2 and 4 by tests/test_dist_cpu.py.  `exact_tail` elements at the end of the flat buffer (the shard-size slot of yolo_amd/train.py: an integer
    that trainer.step divides by) always travel in fp32.  The reference's KVStore reduce is fp32 (car/YOLO.py:160,396):
    'f32' stays the default."""

    def __init__(self, flat, names, offsets, sizes, nbuckets=4, dtype='f32', exact_tail=0):
        if dtype not in ('f32', 'bf16'):
            raise ValueError("GradBuckets dtype must be 'f32' or 'bf16'")
        self.dtype, self.exact_tail = dtype, int(exact_tail)
        self.flat = flat
        self.ranges = bucket_ranges(offsets, sizes, nbuckets, flat.numel())
        self.owner = {}
        for n, o in zip(names, offsets):
            self.owner[n] = next(k for k, (a, b) in enumerate(self.ranges) if a <= o < b)
        self.count0 = [0] * len(self.ranges)
        for n in names:
            self.count0[self.owner[n]] += 1
        # bf16 exchange: what bucket k sends -- [a, b) minus the exact tail -- and its staging buffer (allocated on first use)
        self.stage = [None] * len(self.ranges)
        self.reset()

    def _launch(self, k):
        a, b = self.ranges[k]
        self.launched[k] = True
        if self.dtype == 'f32':
            self.works.append((dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True), None))
            return
        cut = min(b, self.flat.numel() - self.exact_tail)
        if cut > a:
            if self.stage[k] is None:
                self.stage[k] = torch.empty(cut - a, dtype=torch.bfloat16, device=self.flat.device)
            self.stage[k].copy_(self.flat[a:cut])               # (fp32 -> bf16, round-to-nearest-even, on the current stream)
            self.works.append((dist.all_reduce(self.stage[k], op=dist.ReduceOp.SUM, async_op=True), (k, a, cut)))
        if cut < b:
            self.works.append((dist.all_reduce(self.flat[max(cut, a):b], op=dist.ReduceOp.SUM, async_op=True), None))

    def active(self):
        # (YOLO_BENCH_FORCE_DIST exercises the exchange on a single rank)
        return self.enabled and dist.is_initialized() and (dist.get_world_size() > 1 or
                                                           bool(os.environ.get('YOLO_BENCH_FORCE_DIST')))

    def reset(self, enabled=True):
        self.enabled = enabled
        self.pending = list(self.count0)
        self.works = []
        self.launched = [False] * len(self.ranges)

    def done(self, names, before_launch=None):
        """before_launch: called once, right before the first all-reduce these names complete (the caller makes the current
        stream wait for work of another stream that wrote the bucket: the collective is ordered behind the CURRENT stream)."""
        if not self.active():
            return
        for n in names:
            k = self.owner[n]
            self.pending[k] -= 1
            if self.pending[k] == 0 and not self.launched[k]:
                if before_launch is not None:
                    before_launch()
                    before_launch = None
                self._launch(k)

    def wait(self):
        if not self.active():
            return
        for k, (a, b) in enumerate(self.ranges):          # anything the backward never reported (defensive)
            if not self.launched[k]:
                self._launch(k)
        for w, back in self.works:
            w.wait()
            if back is not None:                                # the bf16 sum back into the fp32 gradients
                k, a, cut = back
                self.flat[a:cut].copy_(self.stage[k])
        self.works = []
