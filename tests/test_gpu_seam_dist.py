"""The train-mode network seam (`self.net(bx)` under autograd.record / `.backward()` / `trainer.step(batch_size)`,
car/YOLO.py:381,394,396) and the N > 1 training exchange (configs[3]; car/YOLO.py:372-396, yolo_gluon.py:100-124) on the ONE
GPU a test box has:

  * an EXTERNAL loss (the oracle's torch restatement of _get_loss) on forward(x, training=True)'s logits, handed back through
    backward(grads), must give the gradients of the fused train_step and of the oracle's autograd;
  * the bucketed exchange wired into the backward pass (side-stream weight gradients -> event -> bucket all-reduce ->
    wait -> Adam), run (a) through RCCL with a world of one rank and (b) with a stand-in collective on its own stream that
    doubles the bucket (= two ranks holding the same gradient): an all-reduce launched before a side-stream weight gradient
    of its bucket has landed would leave that parameter at 1x;
  * two shards of a global batch of 4 on the same weights, gradients summed by hand, one Adam step -- against the oracle
    running the two shards with PER-SHARD BatchNorm statistics, summing and rescaling by 1/4;
  * the global batch taken on the device from the slot the exchange reduces = the host-scalar update, bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import graph as og, train as ot, detect as od

pytestmark = pytest.mark.gpu


def _micro(cuda, B=2, dtype='f32', seed_lab=1):
    from yolo_amd.net import CarNet
    spec, size = og.spec_micro(), (64, 96)
    g = og.build_graph(spec)
    P = og.init_params(g, seed=0, bn='random')
    x = np.random.default_rng(2).random((B, 3) + size, dtype=np.float32)
    lab = ot.synthetic_labels(B, seed=seed_lab, render_rate=0.0, num_class=4)
    net = CarNet(spec, dtype=dtype, device=cuda).load_params(P)
    return spec, size, g, P, x, lab, net


def _d53(cuda, B, dtype, tune='auto'):
    from yolo_amd.net import CarNet
    spec, size = og.spec_d53(), (416, 416)
    g = og.build_graph(spec)
    P = og.init_params(g, seed=0, bn='random')
    x = np.random.default_rng(2).random((B, 3) + size, dtype=np.float32)
    lab = ot.synthetic_labels(B, seed=3, render_rate=0.0, num_class=24)
    net = CarNet(spec, dtype=dtype, device=cuda, tune=tune).load_params(P)
    return spec, size, g, P, x, lab, net


def _external_loss(outs, lab, spec, size):
    """The caller's side of car/YOLO.py:381-394 with the ORACLE as the caller: logits (device) -> the five losses of
    _get_loss on the CPU in torch autograd -> d(sum)/d(logits) per scale."""
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    ltrb = od.get_default_ltrb(size, steps, spec['all_anchors'])
    sp = spec['slice_point']
    y, mask = ot.loss_mask(lab, ltrb, spec['all_anchors'], size, steps, area, sp[-1] - sp[-2])
    leaves = [o.detach().cpu().clone().requires_grad_(True) for o in outs]
    merged = torch.cat(leaves, dim=1)
    xs, i = [], 0
    for pt in sp:
        xs.append(merged[..., i:pt]); i = pt
    losses = ot.get_loss(xs, y, ot.score_weight(mask), mask)
    sum(l.sum() for l in losses).backward()
    return [l.detach().numpy() for l in losses], [t.grad for t in leaves]


def _l2(a, b):
    a, b = a.double().flatten().cpu(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_train_mode_seam_external_loss(cuda):
    """forward(x, training=True) -> the caller's own loss -> backward(grads) -> trainer.step(batch)."""
    spec, size, g, P, x, lab, net = _micro(cuda)
    xt = torch.from_numpy(x).to(cuda)
    outs = net.forward(xt, training=True)                                     # builds the default Trainer
    assert [tuple(o.shape) for o in outs] == [(2, h * w, 3, 10) for h, w in ((8, 12), (4, 6), (2, 3))]
    assert all(o.dtype == torch.float32 and o.is_cuda for o in outs)
    rl, rg, rmerged = ot.train_step_reference(g, P, x, lab, spec, size)
    np.testing.assert_allclose(torch.cat(outs, dim=1).cpu().numpy(), rmerged, rtol=0, atol=1e-4 * np.abs(rmerged).max())
    losses, dl = _external_loss(outs, lab, spec, size)
    np.testing.assert_allclose(np.stack(losses), np.stack(rl), rtol=1e-3, atol=1e-7)
    net.backward([d.to(cuda) for d in dl])
    seam = {n: v.clone() for n, v in net.grads().items()}
    assert set(seam) == set(rg)
    # the fused step of the same trainer (device-side targets / losses) from the same state: same gradients
    tr = net.trainer()
    net.load_params(P)                                                        # (the forward moved the running statistics)
    tr.train_step(xt, torch.from_numpy(lab).to(cuda), update=False)
    for n in seam:
        assert _l2(seam[n], tr.grads()[n].cpu()) < 2e-4, n                     # (atomics: summation order differs run to run)
    rel = {n: _l2(seam[n], rg[n]) for n in rg}
    assert np.median(list(rel.values())) < 2e-3 and max(rel.values()) < 0.1, max((v, k) for k, v in rel.items())
    assert max(rel[n] for n in rel if '.out.' in n) < 1e-3
    # the merged-tensor form of backward() is the same thing
    net.load_params(P)
    outs = net(xt, training=True)
    net.backward(torch.cat([d for d in dl], dim=1).to(cuda))
    for n in seam:
        assert _l2(net.grads()[n], seam[n].cpu()) < 2e-4, n
    # trainer.step(batch_size): MXNet Adam on the seam's gradients
    before = {n: net.params[n].clone() for n in seam}
    gnow = {n: v.clone() for n, v in net.grads().items()}
    tr.step(2)
    for n in ('stem.weight', 'heads.0.out.bias', 'stages.2.res.0.c2.gamma'):
        w = before[n].cpu().numpy().copy(); m = np.zeros_like(w); v = np.zeros_like(w)
        ot.adam_step(w, gnow[n].cpu().numpy(), m, v, 1, lr=1e-3, rescale=0.5)
        np.testing.assert_allclose(net.params[n].cpu().numpy(), w, rtol=1e-5, atol=2e-7, err_msg=n)
    # and inference afterwards sees the update
    assert bool(torch.isfinite(net(xt)[0]).all())


def test_seam_misuse_is_loud(cuda):
    from yolo_amd import lib as L
    from yolo_amd.net import CarNet
    spec, size, g, P, x, lab, net = _micro(cuda)
    with pytest.raises(L.YoloError):
        net.backward([torch.zeros(1, device=cuda)])                           # no trainer / no forward yet
    outs = net(torch.from_numpy(x).to(cuda), training=True)
    with pytest.raises(ValueError):
        net.backward([torch.zeros_like(outs[0])])                             # one scale only


@pytest.mark.parametrize('which', ['micro', 'd53'])
def test_two_shards_summed_by_hand_vs_per_shard_bn_oracle(cuda, which):
    """configs[3] emulated on one GPU (fp32; the micro net with tight bars, and the D53 spec at 416x416 -- configs[3]'s own
    graph -- with the whole-step bars of tests/test_gpu_configs.py): a global batch of 4 = two ranks x 2 images.  Each 'rank' runs its shard on
    the same weights with its OWN batch statistics (no SyncBN, car/YOLO.py:94-96), the gradient buffers are summed as the
    all-reduce would, then trainer.step(4).  Oracle: the same two shard passes in torch autograd, summed, Adam with
    rescale 1/4 (car/YOLO.py:372-396)."""
    from yolo_amd.train import Trainer
    if which == 'micro':
        spec, size, g, P, x, lab, net = _micro(cuda, B=4, seed_lab=7)
        bar_med, bar_max = 2e-3, 0.1
    else:
        spec, size, g, P, x, lab, net = _d53(cuda, 4, 'f32')
        bar_med, bar_max = 2e-2, 0.15
    tr = Trainer(net, size)
    w0 = tr.wflat.clone()
    shard_g, shard_l, ref_g, ref_l = [], [], [], []
    for r in range(2):
        a, b = int(r * 4 / 2), int((r + 1) * 4 / 2)                           # yolo_gluon.py:118-119
        net.load_params(P)
        l = tr.train_step(torch.from_numpy(x[a:b]).to(cuda), torch.from_numpy(lab[a:b]).to(cuda), update=False)
        shard_g.append(tr.gflat.clone()); shard_l.append(l.cpu().numpy())
        assert float(tr.gflat[tr.nparam]) == 2.0                              # the shard size rides in the buffer's slot
        rl, rg, _ = ot.train_step_reference(g, P, x[a:b], lab[a:b], spec, size)
        ref_g.append(rg); ref_l.append(np.stack(rl))
    np.testing.assert_allclose(np.concatenate(shard_l, axis=1), np.concatenate(ref_l, axis=1), rtol=1e-3, atol=1e-7)
    # whole-batch statistics would give other losses: the shards really normalise on their own
    whole, _, _ = ot.train_step_reference(g, P, x, lab, spec, size)
    assert np.abs(np.stack(whole) - np.concatenate(ref_l, axis=1)).max() > 1e-4
    net.load_params(P)
    assert bool((tr.wflat == w0).all())
    tr.gflat.copy_(shard_g[0] + shard_g[1])                                   # the SUM all-reduce, by hand
    assert float(tr.gflat[tr.nparam]) == 4.0                                  # ... which also delivers the global batch
    summed = {n: v.clone() for n, v in tr.grads().items()}
    tr.step(4)
    rel = {}
    for n in summed:
        rsum = ref_g[0][n] + ref_g[1][n]
        rel[n] = _l2(summed[n], rsum)
        # Adam, exactly: the oracle's formula on the very gradient the HIP path summed
        w = P[n].copy(); m = np.zeros_like(w); v = np.zeros_like(w)
        ot.adam_step(w, summed[n].cpu().numpy(), m, v, 1, lr=1e-3, rescale=0.25)
        np.testing.assert_allclose(net.params[n].cpu().numpy(), w, rtol=1e-5, atol=2e-7, err_msg=n)
    assert max(rel[n] for n in rel if '.out.' in n) < 1e-3
    assert np.median(list(rel.values())) < bar_med and max(rel.values()) < bar_max, max((v, k) for k, v in rel.items())
    # end to end: the oracle's own summed gradient through the oracle's Adam lands where the HIP weights are
    tight = [n for n in rel if rel[n] < 1e-3]
    assert len(tight) > (len(rel) // 2 if which == 'micro' else 5)
    for n in tight:
        w = P[n].copy(); m = np.zeros_like(w); v = np.zeros_like(w)
        ot.adam_step(w, ref_g[0][n] + ref_g[1][n], m, v, 1, lr=1e-3, rescale=0.25)
        # (Adam's first step is lr * sign(g) wherever |g| >> eps: compare the step taken, not the weight)
        d_hip = net.params[n].cpu().numpy() - P[n]
        d_ref = w - P[n]
        assert np.abs(d_hip - d_ref).mean() < 0.05 * np.abs(d_ref).mean() + 1e-9, n


class _FakeWork(object):
    def __init__(self, stream):
        self.ev = torch.cuda.Event()
        self.ev.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def _doubling_all_reduce(comm):
    """Stand-in for dist.all_reduce(SUM) of two ranks that hold the same gradient, with RCCL's stream semantics: the
    collective is ordered behind the stream that is current at the call and runs on the communicator's OWN stream; the
    returned work's wait() orders the current stream behind it.  The weight gradients accumulate atomically into the zeroed
    buffer: one that lands after its bucket was doubled leaves that parameter at 1x."""
    def all_reduce(t, op=None, async_op=False):
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            t.mul_(2.0)
        w = _FakeWork(comm)
        if not async_op:
            w.wait()
        return w
    return all_reduce


def test_bucketed_exchange_ordering_with_a_stand_in_collective(cuda, monkeypatch):
    """D53 416x416 B=4 bf16, measured variants (the weight gradients run on the side stream, 4 buckets): with the stand-in
    collective every gradient must come out at exactly 2x the local one -- never 1x (reduced before the side stream
    wrote it) and never 4x -- and the update must use global batch 2 x 4 = 8 from the reduced slot."""
    from yolo_amd import parallel
    from yolo_amd.train import Trainer
    import torch.distributed as dist
    spec, size, g, P, x, lab, net = _d53(cuda, 4, 'bf16', tune='measure')
    tr = Trainer(net, size)
    xt, lt = torch.from_numpy(x).to(cuda), torch.from_numpy(lab).to(cuda)
    tr.train_step(xt, lt, update=False)                                       # (measures the variants)
    net.load_params(P)
    tr.train_step(xt, lt, update=False)
    local = tr.gflat.clone()
    comm = torch.cuda.Stream(device=cuda)
    monkeypatch.setattr(dist, 'all_reduce', _doubling_all_reduce(comm))
    monkeypatch.setattr(parallel.GradBuckets, 'active', lambda self: self.enabled)
    for rep in range(3):
        net.load_params(P)
        tr.train_step(xt, lt, update=False)                                   # exchange off: this repetition's reference
        ref = tr.gflat.clone()
        net.load_params(P)
        tr.forward(xt)
        P_ = tr._plans[4]
        tr._backward(P_, exchange=True)                                       # dmerged still holds the step's dlogits
        assert all(tr.buckets.launched) and len(tr.buckets.works) == len(tr.buckets.ranges) >= 2
        tr.buckets.wait()
        torch.cuda.synchronize()
        got = tr.gflat.clone()
        assert float(got[tr.nparam]) == 8.0
        for n in tr.names:
            a, b = tr.gview[n], None
            o = a.data_ptr() - tr.gflat.data_ptr()
            sl = slice(o // 4, o // 4 + a.numel())
            r2 = 2.0 * ref[sl].double()
            err = float((got[sl].double() - r2).norm() / (r2.norm() + 1e-30))
            assert err < 1e-3, 'rep %d %s: |got - 2 x local| / |2 x local| = %.3g (1x would read 0.5)' % (rep, n, err)
    # the optimiser reads 1/8 from the slot
    w_before = tr.wflat.clone()
    tr.t = 0; tr.mflat.zero_(); tr.vflat.zero_()
    tr.step()
    w_dev = tr.wflat.clone()
    tr.wflat.copy_(w_before); tr.t = 0; tr.mflat.zero_(); tr.vflat.zero_()
    tr.gflat.copy_(got)
    tr.step(8)
    assert bool((tr.wflat[:tr.nparam] == w_dev[:tr.nparam]).all())
    assert float((local - ref).norm() / ref.norm()) < 1e-2                     # (bf16 step, atomics: same gradient up to order)


def test_rccl_world_of_one_exchange_is_the_identity(cuda):
    """The real thing with the one rank a test box has: torch.distributed 'nccl' (= RCCL), world size 1,
    YOLO_BENCH_FORCE_DIST=1 makes the Trainer run its bucketed exchange: 4 asynchronous all-reduces launched from inside
    the backward pass, wait() before Adam, the global batch from the reduced slot.  A SUM over one rank is the identity:
    gradients equal the exchange-off gradients (up to the run-to-run order of the weight gradients' atomics) and the
    updated weights are BIT-identical to the host-scalar update of the same gradient buffer."""
    import torch.distributed as dist
    from yolo_amd.train import Trainer
    import socket
    so = socket.socket(); so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]; so.close()
    saved = {k: os.environ.get(k) for k in ('MASTER_ADDR', 'MASTER_PORT', 'YOLO_BENCH_FORCE_DIST')}
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), YOLO_BENCH_FORCE_DIST='1')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=cuda)
    try:
        spec, size, g, P, x, lab, net = _d53(cuda, 4, 'bf16', tune='measure')
        tr = Trainer(net, size)
        xt, lt = torch.from_numpy(x).to(cuda), torch.from_numpy(lab).to(cuda)
        tr.train_step(xt, lt, update=False)                                   # measure the variants; exchange off
        net.load_params(P)
        tr.train_step(xt, lt, update=False)
        off = tr.gflat.clone()
        assert not tr.buckets.works and not any(tr.buckets.launched)
        net.load_params(P)
        tr.forward(xt)
        tr._backward(tr._plans[4], exchange=True)                             # same dlogits, exchange on
        assert tr.buckets.active() and all(tr.buckets.launched) and len(tr.buckets.works) >= 2
        tr.buckets.wait()
        on = tr.gflat.clone()
        assert float(on[tr.nparam]) == 4.0
        rel = float((on - off).double().norm() / off.double().norm())
        assert rel < 1e-2, rel
        worst = 0.0
        for n in tr.names:
            o = (tr.gview[n].data_ptr() - tr.gflat.data_ptr()) // 4
            sl = slice(o, o + tr.gview[n].numel())
            worst = max(worst, float((on[sl] - off[sl]).double().norm() / (off[sl].double().norm() + 1e-30)))
        print('exchange on vs off, worst parameter: %.3g' % worst)
        # update through the exchange (device slot) vs the host scalar on the same buffer: bit-identical weights
        w0 = tr.wflat.clone()
        tr.step()
        w_dev = tr.wflat.clone()
        tr.wflat.copy_(w0); tr.t = 0; tr.mflat.zero_(); tr.vflat.zero_(); tr.gflat.copy_(on)
        tr.buckets.reset(enabled=False)
        tr.step(4)
        assert bool((tr.wflat[:tr.nparam] == w_dev[:tr.nparam]).all())
        # and three whole steps through the public call, exchange on: finite, decreasing-ish, no hang
        net.load_params(P)
        ls = [float(tr.train_step(xt, lt).sum()) for _ in range(3)]
        assert all(np.isfinite(ls)), ls
    finally:
        dist.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# ---- two REAL ranks (two processes, torch.distributed) on the one GPU of a test box --------------------------------------
def _two_rank_worker(rank, world, port, q):
    """One rank of a 2-rank job on cuda:0 (gloo carries CUDA tensors through the host: slow, but a real cross-process
    SUM all-reduce with the stream semantics of an asynchronous collective)."""
    import torch.distributed as dist
    from yolo_amd.net import CarNet
    from yolo_amd.train import Trainer
    from yolo_amd import parallel
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cuda = torch.device('cuda:0')
        spec, size = og.spec_micro(), (64, 96)
        P = og.init_params(og.build_graph(spec), seed=0, bn='random')
        # a global batch of 5 over 2 ranks: UNEVEN shards 2 + 3 (yolo_gluon.py:118-119)
        x = np.random.default_rng(2).random((5, 3) + size, dtype=np.float32)
        lab = ot.synthetic_labels(5, seed=7, render_rate=0.0, num_class=4)
        a, b = parallel.shard_bounds(5, rank, world)
        net = CarNet(spec, dtype='f32', device=cuda).load_params(P)
        tr = Trainer(net, size)
        assert tr.buckets.active()
        losses = tr.train_step(torch.from_numpy(x[a:b]).to(cuda), torch.from_numpy(lab[a:b]).to(cuda))     # exchange + Adam
        torch.cuda.synchronize()
        q.put((rank, b - a, float(tr.gflat[tr.nparam]), losses.cpu().numpy(), tr.gflat.cpu().numpy(), tr.wflat.cpu().numpy()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_real_ranks_exchange_on_one_gpu(cuda):
    """configs[3] with two PROCESSES: each rank runs its uneven shard (2 + 3 images) with its own batch statistics, the
    Trainer's bucketed asynchronous all-reduce sums the gradient buffers between the processes, the global batch (5) arrives
    in the buffer's slot, every rank applies the same Adam step.  Against the single-process emulation of the same thing:
    the two shards run one after the other on one Trainer, the buffers added by hand, step(5)."""
    import torch.multiprocessing as mp
    import socket
    from yolo_amd.net import CarNet
    from yolo_amd.train import Trainer
    so = socket.socket(); so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]; so.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    [p.join(timeout=120) for p in ps]
    assert [o[1] for o in out] == [2, 3] and out[0][2] == out[1][2] == 5.0          # the slot: SUM of the shard sizes
    np.testing.assert_array_equal(out[0][4], out[1][4])                                # the same reduced gradient buffer ...
    np.testing.assert_array_equal(out[0][5], out[1][5])                                # ... and the same weights on both ranks
    # single-process emulation
    spec, size = og.spec_micro(), (64, 96)
    P = og.init_params(og.build_graph(spec), seed=0, bn='random')
    x = np.random.default_rng(2).random((5, 3) + size, dtype=np.float32)
    lab = ot.synthetic_labels(5, seed=7, render_rate=0.0, num_class=4)
    net = CarNet(spec, dtype='f32', device=cuda).load_params(P)
    tr = Trainer(net, size)
    gsum, ls = None, []
    for a, b in ((0, 2), (2, 5)):
        net.load_params(P)
        ls.append(tr.train_step(torch.from_numpy(x[a:b]).to(cuda), torch.from_numpy(lab[a:b]).to(cuda), update=False).cpu().numpy())
        gsum = tr.gflat.clone() if gsum is None else gsum + tr.gflat
    net.load_params(P)
    tr.gflat.copy_(gsum)
    tr.step(5)
    np.testing.assert_allclose(np.concatenate([out[0][3], out[1][3]], axis=1), np.concatenate(ls, axis=1), rtol=1e-5, atol=1e-8)
    g_ref, g_got = gsum.cpu().numpy(), out[0][4]
    assert np.abs(g_got - g_ref).max() <= 1e-5 * np.abs(g_ref).max() + 1e-9         # (atomics: summation order inside a shard)
    w_ref, w_got = tr.wflat.cpu().numpy()[:tr.nparam], out[0][5][:tr.nparam]
    # Adam's first step is lr * g / (|g| + eps): weights agree wherever the two gradients do
    assert np.mean(np.abs(w_got - w_ref) > 1e-6) < 1e-3, float(np.mean(np.abs(w_got - w_ref) > 1e-6))


@pytest.mark.parametrize('world', [2, 8])
def test_bench_two_ranks_sharing_the_gpu(world):
    """`bench.py --gpus N` (N = 2, and N = 8 = BASELINE configs[3] / [4]'s world) end to end on a one-GPU box (TEST-ONLY switch
    YOLO_BENCH_SHARED_GPU + gloo): the launcher starts N ranks, all run the sharded inference passes and the training pass with
    the real exchange, rank 0 prints ONE JSON line last on stdout with n_gpus = N, the training key with its exchange block,
    and the test-only marker."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(YOLO_BENCH_BACKEND='gloo', YOLO_BENCH_SHARED_GPU='1')
    import gc
    gc.collect()
    torch.cuda.empty_cache()                # (the N ranks share THIS process's GPU: give back what the earlier tests cached)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', '3', '--warmup', '1', '--no-northstar',
                        '--no-roofline', '--no-repeats', '--sustain-steps', '0', '--train-timeout', '900'], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                                          # the JSON line is the LAST line
    assert d['n_gpus'] == world and d['config']['global_batch'] == 32 * world and 'shared_gpu_test' in d
    t = d['train_416_bs64']
    assert 'error' not in t, t
    # BASELINE configs[3]: the training loop at GLOBAL batch 256 sharded over the ranks (8 x 32); the 64-per-GPU weak-scaling pass
    # is an extra key
    assert t['global_batch'] == 256 and t['batch_per_gpu'] == 256 // world
    assert d['train_416_bs64_weak']['global_batch'] == 64 * world and d['train_416_bs64_weak']['batch_per_gpu'] == 64
    assert t['n_gpus'] == world and t['exchange']['rccl_world'] == world and t['exchange']['buckets'] >= 2
    assert t['exchange']['allreduce_ms_per_step_standalone'] > 0 and all(np.isfinite(t['final_losses']))
    assert 'cpu_baseline' not in d and 'f32_path' not in d                              # rank-0-only extras of the N = 1 line
    # rank 0 measured the kernel variants, rank 1 adopted its choices: the same launch plan and the same training-step
    # variants on both ranks (tune='measure' is box- and rank-dependent when left alone), and every rank's own time is in the line
    assert d['plans_identical_across_ranks'] is True and t['tuning_identical_across_ranks'] is True
    assert len(d['per_rank_ms_per_step']) == world and len(t['per_rank_ms_per_step']) == world and 'errors' not in d


def test_bench_line_launches_the_committed_plan_and_quotes_its_own_pmc_pass():
    """bench.py's default command (what the driver runs) launches profiles/plan.json: no shape is measured live, and the launch
    plan it arrives at is the one the committed PMC pass was taken with -- `roofline.traffic` is quoted (a summary of another
    plan would be refused: the lookup is keyed by plan_md5 and launches per step).  Guards profiles/ against a plan that was
    re-made without refreshing the passes, and the plan against kernels that were renamed without re-making it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-northstar', '--no-train-key',
                        '--no-f32-key', '--no-cpu-baseline', '--no-repeats'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d['plan']['mode'] == 'plan' and d['plan']['file'] == 'profiles/plan.json' and d['plan']['measured_live'] == 0, d['plan']
    rf = d['roofline']
    assert rf['traffic'] is not None and rf['traffic_source'].startswith('profiles/'), rf
    assert rf['traffic'] >= 0.9 * rf['algorithmic_bytes'] and 0.2 < rf['frac'] < 1.0
    assert d['dtype'] == 'bf16' and d['config']['global_batch'] == 32 and d['unit'] == 'images/s'
