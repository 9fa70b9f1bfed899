#!/usr/bin/env python
"""bench.py -- images/s of the YOLOv3 hot path on MI355X (BASELINE.json metric / configs[1]).

One step = one pass of the hot path over one batch of synthetic images already resident in HBM:
NCHW float32 images -> Darknet-53-spec backbone + 3-scale YOLO heads (bf16 MFMA, fp32 accumulate)
-> anchor decode + per-image top-1 (the reference's `predict`, car/YOLO.py:568-597), all on device.
Batch is sharded across ranks (one process per GPU, weak scaling, no data-path collective:
SURVEY.md section 8e "inference").

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is for the kernel instantiation with the largest
total time: achieved = algorithmic conv FLOPs of its launches / their HIP-event durations,
measured live in a second pass over the same K steps (events bracket each launch on the stream the
kernels run on; the headline `value` pass itself carries no events).  `cpu_baseline` times the
CPU oracle (torch-CPU fp32 restatement of the reference graph; the reference's MXNet cannot run
here) on a bounded sample on rank 0 at N=1.

The step OWNS its device->host copy, as the reference's `predict` does (car/YOLO.py:597): the (B, 6+C) rows go to a
pinned host buffer by an asynchronous copy on the stream, two buffers deep, so the copy of step i overlaps the launches
of step i+1 (`value_blocking_predict` is the same pass with a blocking `.cpu()` per step).  The extra keys on the 608x608
bs 64 shape (`northstar_608_forward` / `northstar_608` / `northstar_608_nms`) and the training key are SUSTAINED runs:
>= 2 s of the same pass as pre-heat, >= 20 warm-up steps, then >= 300 (training: 120) timed steps, with the mean shader
clock and socket power sampled over the timed region (`sustained_s`, `sclk_mhz`, `power_w`).  The headline `value`
keeps the driver's --steps / --warmup exactly.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3,       # /opt/skills/guides/MI355X_MICROARCH.md (dense)
                    'bf16x3': 2500.0 / 3, 'f16x3': 2500.0 / 3}      # split bf16 / f16: three 2-byte MFMAs per algorithmic product


class Telemetry(object):
    """Mean shader clock (MHz) and socket power (W) over a timed region, sampled from a background thread: the amdgpu
    hwmon files when they exist (a read costs microseconds), else `rocm-smi --showclocks --showpower` once a second
    (what tools/clock_probe.sh does).  Purely descriptive: nothing in the timed region waits for it."""

    def __init__(self, index=0, period=0.2):
        import glob
        self.period, self.index = period, index
        self.freq_file = self.power_file = None
        # the hwmon directory of THIS device: by its PCI address (a box lists more DRM cards than the GPUs a job may see)
        hw = None
        try:
            pr = torch.cuda.get_device_properties(index)
            bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            found = sorted(glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % bdf))
            hw = found[0] if found else None
        except Exception:
            hw = None
        if hw is not None:
            if os.path.exists(os.path.join(hw, 'freq1_input')):
                self.freq_file = os.path.join(hw, 'freq1_input')
            for name in ('power1_average', 'power1_input'):
                if os.path.exists(os.path.join(hw, name)):
                    self.power_file = os.path.join(hw, name)
                    break
        self.source = 'hwmon' if (self.freq_file or self.power_file) else 'rocm-smi'
        self._stop = self._thread = None
        self.clk, self.pw = [], []

    def _sample(self):
        import re
        import subprocess
        if self.source == 'hwmon':
            try:
                if self.freq_file:
                    self.clk.append(int(open(self.freq_file).read()) / 1e6)
                if self.power_file:
                    self.pw.append(int(open(self.power_file).read()) / 1e6)
            except (OSError, ValueError):
                pass
            return
        try:
            txt = subprocess.run(['rocm-smi', '-d', str(self.index), '--showclocks', '--showpower'], capture_output=True, text=True,
                                 timeout=10).stdout
        except Exception:
            return
        m = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', txt)
        if m:
            self.clk.append(float(m.group(1)))
        m = re.search(r'Power \(W\): ([0-9.]+)', txt)
        if m:
            self.pw.append(float(m.group(1)))

    def start(self):
        import threading
        self.clk, self.pw = [], []
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                self._sample()
                self._stop.wait(self.period if self.source == 'hwmon' else 1.0)

        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=15)
        mean = lambda v: round(sum(v) / len(v), 1) if v else None
        return {'sclk_mhz': mean(self.clk), 'power_w': mean(self.pw), 'telemetry_samples': max(len(self.clk), len(self.pw)),
                'telemetry_source': self.source}


def host_topology():
    """Physical layout of the host CPUs this process may run on, from /sys: {socket id: [one hardware thread per physical core]}
    (the first SMT sibling of every core).  Falls back to one 'socket' of all allowed CPUs when /sys says nothing."""
    allowed = sorted(os.sched_getaffinity(0))
    socks, seen = {}, set()
    for c in allowed:
        base = '/sys/devices/system/cpu/cpu%d/topology/' % c
        try:
            pkg = int(open(base + 'physical_package_id').read())
            core = int(open(base + 'core_id').read())
        except (OSError, ValueError):
            return {0: allowed}
        if (pkg, core) not in seen:
            seen.add((pkg, core))
            socks.setdefault(pkg, []).append(c)
    return socks or {0: allowed}


def cpu_quota():
    """CPUs the container may use per scheduling period (cgroup v2 cpu.max, v1 cpu.cfs_quota_us), or None when unlimited."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_child(cfg):
    """The timed CPU leg in a process of its own: affinity and the OpenMP environment must be in place BEFORE torch creates its
    thread pool (a pool created earlier keeps the whole-machine affinity whatever sched_setaffinity says afterwards) -- the parent
    starts this process already pinned (preexec_fn) with OMP_PROC_BIND / OMP_PLACES set.  Prints one JSON line.
    cfg: {cpus, size, batch, seconds}."""
    os.sched_setaffinity(0, cfg['cpus'])
    t = torch
    from oracle import graph as og, forward as of, detect as od
    t.set_num_threads(len(cfg['cpus']))
    size, batch = tuple(cfg['size']), cfg['batch']
    spec = og.spec_d53()
    g = og.build_graph(spec)
    Pt = {k: t.from_numpy(v) for k, v in og.init_params(g, seed=0, bn='identity').items()}
    x = t.rand((batch, 3) + size)
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])

    def once():
        with t.no_grad():
            outs = of.forward_torch(g, Pt, x)
        od.predict([o.numpy() for o in outs], spec['slice_point'], size, syxhw)

    once()                                                # warm-up (allocations, oneDNN primitive cache)
    n, t0 = 0, time.time()
    while True:
        once(); n += 1
        el = time.time() - t0
        if el >= cfg['seconds'] or n >= 10:
            break
    print(json.dumps({'img_s': n * batch / el, 'iters': n, 'elapsed_s': el, 'threads': t.get_num_threads()}))


def cpu_baseline(size, seconds=6.0, batch=32, dev=None, plan_state=None, plan_batch=None):
    """Oracle forward + numpy decode/top-1 on the host cores: images/s on a bounded sample (batch 32: the headline's own batch).

    (round 6) The host is USED, not just present: each candidate runs in a child process pinned (sched_setaffinity before torch's
    thread pool exists, OMP_PROC_BIND=close) to one hardware thread per PHYSICAL core -- of one socket, and of all sockets -- and the
    best is reported with its core and socket count and its GFLOP/s (round 5 let oneDNN spread 32-64 unpinned threads over a
    256-thread two-socket host: 4.5 img/s, what the 8-core build container reaches).

    dev: the oracle also CHECKS the HIP path here (it is the checker, never the thing measured): the first two images of the
    sample through CarNet in every arithmetic path, decoded, against the oracle's rows -> `box_parity` (max / RMS of
    |a - b| / (1 + |b|) over [l, t, r, b] of all boxes, top-1 index agreement): the `north_star` tolerance as a number in the line."""
    import subprocess
    from oracle import graph as og, forward as of, detect as od
    spec = og.spec_d53()
    g = og.build_graph(spec)
    P = og.init_params(g, seed=0, bn='identity')
    Pt = {k: torch.from_numpy(v) for k, v in P.items()}
    x = torch.rand((2, 3) + tuple(size))
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])
    topo = host_topology()
    socks = sorted(topo)
    quota = cpu_quota()
    s0 = topo[socks[0]]
    # placements, all one hardware thread per physical core of socket 0 (+ every socket's cores where nothing limits the process):
    # as many cores as the container's CPU quota pays for -- more threads than that are throttled by the scheduler (the pool's
    # boxes: cpu.max = 16 CPUs of a 2 x 64-core host: 64 pinned threads ran at 1.7 img/s, 32 at 3.7) -- and half of that
    cands = []
    if quota is not None and quota < len(s0):
        n = max(1, int(quota))
        cands += [('%d cores of 1 socket (the container\'s CPU quota)' % n, s0[:n], 1)]
        if n >= 8:
            cands += [('%d cores of 1 socket' % (n // 2), s0[:n // 2], 1)]
        cands += [('%d cores of 1 socket' % min(2 * n, len(s0)), s0[:min(2 * n, len(s0))], 1)]
    else:
        cands += [('1 socket', s0, 1)]
        if len(socks) > 1:
            cands.append(('%d sockets' % len(socks), [c for s_ in socks for c in topo[s_]], len(socks)))
        if len(s0) >= 16:
            cands.append(('half a socket', s0[:len(s0) // 2], 1))
    tried, best = {}, None
    gflop = og.conv_flops(g, *size) / 1e9 if hasattr(og, 'conv_flops') else None
    for label, cpus, nsock in cands:
        env = dict(os.environ, OMP_NUM_THREADS=str(len(cpus)), OMP_PROC_BIND='close', OMP_PLACES='cores', MKL_NUM_THREADS=str(len(cpus)))
        cfg = {'cpus': cpus, 'size': list(size), 'batch': batch, 'seconds': seconds}
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-child', json.dumps(cfg)], env=env, capture_output=True,
                               text=True, timeout=max(120.0, 20 * seconds), preexec_fn=lambda c_=cpus: os.sched_setaffinity(0, c_))
            res = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:                            # (a candidate that cannot run is reported, not fatal)
            tried[label] = 'failed: %s' % (str(e)[:80],)
            continue
        tried[label] = {'cores': len(cpus), 'img_s': round(res['img_s'], 3), 'iters': res['iters'], 'elapsed_s': round(res['elapsed_s'], 1)}
        if best is None or res['img_s'] > best[0]:
            best = (res['img_s'], len(cpus), res['iters'], res['elapsed_s'], nsock, label)
        elif res['img_s'] < 0.6 * best[0]:
            break                                         # (the placements are ordered most-promising first)
    if best is None:
        raise RuntimeError('cpu_baseline: no candidate ran: %s' % json.dumps(tried))
    parity = None
    if dev is not None:
        from yolo_amd.net import CarNet
        from yolo_amd.detect import Detector
        with torch.no_grad():
            ref = [o.numpy() for o in of.forward_torch(g, Pt, x)]
        ref_rows = od.decode_all(ref, spec['slice_point'], size, syxhw)
        _, ref_idx = od.predict(ref, spec['slice_point'], size, syxhw)
        ref_kept = [[int(v) for v in od.nms(ref_rows[i], mode='class')[0]] for i in range(2)]
        det = Detector(spec, size, steps, device=dev)
        parity = {'sample': 'images 0-1 of the cpu_baseline sample (D53 spec, identity BN, Xavier weights seed 0), decoded rows vs the fp32 oracle',
                  'measure': 'max / RMS of |a - b| / (1 + |b|) over [l,t,r,b] of all %d boxes' % ref_rows.shape[1]}
        # (round 6) the checked nets launch the COMMITTED PLAN's kernels -- the set the headline is made of -- at the plan's batch
        # (the two images repeated: eval-mode images are independent); without a plan the variants are measured on this box
        rep = max(1, (plan_batch or 2) // 2)
        parity['kernels'] = 'committed plan, batch %d (images 0-1 repeated)' % (2 * rep) if plan_state is not None else 'measured on this box'
        for dt in ('f32', 'f16x3', 'bf16x3', 'f16', 'bf16'):
            net = CarNet(spec, dtype=dt, device=dev, tune='measure').load_params(P)
            if plan_state is not None:
                net.load_tuning_state(plan_state)
            outs = net(x.to(dev).repeat(rep, 1, 1, 1))
            rows = det.decode(outs).cpu().numpy()[:2]
            _, idx = det.predict_device(outs)
            idx = idx[:2]
            e = (rows[..., 1:5].astype(np.float64) - ref_rows[..., 1:5]) / (1.0 + np.abs(ref_rows[..., 1:5]))
            # per-class NMS end to end: this path's logits through the HIP decode + NMS against the oracle's logits through the oracle's
            # own decode + NMS (north_star: "kept indices")
            rows_d = det.decode(outs)
            kept, _, cnt = det.nms(rows_d, 'class', scores=det.nms_scores(rows_d, 'class'))
            kept, cnt = kept.cpu().numpy()[:2], cnt.cpu().numpy()[:2]
            same = [[int(v) for v in kept[i, :int(cnt[i])]] == ref_kept[i] for i in range(2)]
            parity[dt] = {'box_max': float(np.abs(e).max()), 'box_rms': float(np.sqrt(np.mean(e * e))),
                          'score_max': float(np.abs(rows[..., 0] - ref_rows[..., 0]).max()),
                          'top1_index_agreement': float(np.mean(idx.cpu().numpy() == ref_idx)),
                          'nms_kept_lists_identical': float(np.mean(same))}
            del net
        torch.cuda.empty_cache()
    flops_img = 113.26e9 * (size[0] * size[1]) / (416.0 * 416.0)          # 113.26 GFLOP per 416x416 image, ~ pixels
    return dict(value=round(best[0], 3), unit='images/s', cores=best[1], sockets=best[4], sockets_in_host=len(socks),
                physical_cores_in_host=sum(len(v) for v in topo.values()), hardware_threads=os.cpu_count(), cpu_quota=quota,
                gflops=round(best[0] * flops_img / 1e9, 1), kind='port', box_parity=parity, tried=tried,
                sample='oracle.forward_torch (torch-CPU fp32 oneDNN restatement of the reference graph; MXNet cannot run '
                       'here) + numpy decode/top-1, D53 spec %dx%d, batch %d x %d iterations after 1 warm-up (%.1f s), in a child '
                       'process pinned to one hardware thread per physical core of %s (%d cores); best of the placements tried'
                       % (size[0], size[1], batch, best[2], best[3], best[5], best[1]))


def pmc_traffic(kernel, B, size, plan_md5, launches_per_step):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 PMC summary (profiles/*_pmc_traffic.json, made by
    tools/pmc_traffic.py from separate --pmc passes: FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, + WRITE_SIZE)
    -- but only from a summary of the SAME launch plan: same workload, same plan_md5 (md5 over the [(op, kernel)] launch list)
    and the same number of launches of this kernel per step.  A kernel NAME alone does not identify the layers it ran (round 4:
    the tuner gave the name 7 layers on one box and 14 on another).  -> (bytes, file, note)"""
    import glob
    seen = []
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
        with open(path) as f:
            prof = json.load(f)
        if prof.get('workload') != [B, size[0], size[1]]:
            continue
        ent = prof.get('kernels', {}).get(kernel)
        rel = os.path.relpath(path, ROOT)
        if prof.get('plan_md5') == plan_md5 and ent and ent.get('launches_per_step') == launches_per_step:
            return int(ent['hbm_bytes_per_launch']), rel, None
        seen.append('%s (plan %s%s)' % (rel, prof.get('plan_md5'), '' if not ent else ', %s launches of this kernel per step'
                                        % ent.get('launches_per_step')))
    return None, None, ('no committed PMC pass describes this launch plan (plan_md5 %s, %d launches of the dominant kernel per step); '
                        'passes of this workload on file: %s' % (plan_md5, launches_per_step, '; '.join(seen[:3]) or 'none'))


def synthetic_labels(batch, seed, num_class=24):
    """Config-3 synthetic targets: (B,1,6+C) rows [cls,y,x,h,w,rot,dist...]; with p = 0.5 a row
    of -1 = no object (render_rate, car/YOLO.py:329); dist = normalised exp(-d^2/0.1) over 24 azimuth bins."""
    rng = np.random.default_rng(seed)
    lab = -np.ones((batch, 1, 6 + num_class), np.float32)
    centres = np.deg2rad(15.0 * np.arange(num_class))
    for b in range(batch):
        if rng.random() < 0.5:
            continue
        azi = rng.uniform(0, 2 * np.pi)
        d = np.arccos(np.clip(np.cos(azi - centres), -1, 1))
        g = np.exp(-d.astype(np.float32) ** 2 / np.float32(0.1))
        y, x = rng.uniform(.15, .85, 2)
        h, w = rng.uniform(.2, .9, 2)
        lab[b, 0, :6] = [int(np.argmin(d)), y, x, h, w, rng.uniform(-30, 30) * np.pi / 180]
        lab[b, 0, 6:] = g / g.sum()
    return lab


def train_pass(args, spec, size, B, rank, world, dev, dist, steps, warmup, preheat_s=0.0, telemetry=None):
    """BASELINE configs[2] (N=1) / configs[3] (N>1): car/YOLO.py training step, B images per GPU, the gradient
    bucket all-reduced over RCCL (one exchange per step), identical Adam on every rank.  Returns the result dict."""
    from yolo_amd.net import CarNet
    from yolo_amd.train import Trainer
    from yolo_amd import parallel
    net = CarNet(spec, dtype=args.dtype, device=dev, tune='measure', tune_cache=args.tune_cache,
                 fuse_stem=not args.no_fuse_stem).initialize(seed=1234)
    tr = Trainer(net, size, grad_exchange=args.grad_exchange, grad_buckets=args.grad_buckets)
    if getattr(args, 'plan_state', None) is not None:
        tr.load_tuning_state(args.plan_state)
    gen = torch.Generator(device='cpu').manual_seed(100 + rank)
    x = torch.rand((B, 3) + size, generator=gen).to(dev)
    lab = torch.from_numpy(synthetic_labels(B, 3 + rank)).to(dev)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    tuning_same = None
    if dist is not None:
        # rank 0 measures the forward / data-gradient / weight-gradient variants in one local step (no exchange, no update),
        # every rank adopts its choices: all ranks run the same kernels, nobody is the straggler the all-reduce waits for
        if rank == 0:
            tr.tune(x, lab)
        parallel.share_tuning(tr, src=0)
        tr.train_step(x, lab)                     # (every rank's first step: plan build with rank 0's choices, one real exchange)
        # (a rank that had to measure a shape itself would hold choices rank 0 never made)
        tuning_same = parallel.same_on_all_ranks(sorted((n, repr(k), repr(v)) for n, d_ in tr.tuning_state().items() for k, v in d_.items()))
    if preheat_s > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < preheat_s:
            for _ in range(3):
                tr.train_step(x, lab)
            torch.cuda.synchronize()
            if dist is not None:                  # (the ranks leave the pre-heat together: a rank still inside would hang the others' exchange)
                go = torch.tensor([1.0 if time.perf_counter() - t0 < preheat_s else 0.0], device=dev)
                dist.all_reduce(go, op=dist.ReduceOp.MIN)
                if float(go.item()) == 0.0:
                    break
    for _ in range(max(warmup, 1)):               # the first step also measures the kernel variants
        tr.train_step(x, lab)
    fence()
    if telemetry is not None:
        telemetry.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.train_step(x, lab)
    fence()
    el = time.perf_counter() - t0
    tinfo = telemetry.stop() if telemetry is not None else {}
    per_rank = [el]
    if dist is not None:
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = el
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [float(v) for v in t.tolist()]
        el = max(per_rank)
    value = world * B * steps / el
    fl = 3 * net.graph.flops(*size)
    tf = fl * value / 1e12 / world
    exch = None
    if dist is not None and tr.buckets.active():
        # the exchange on its own (not overlapped): the 4 bucket all-reduces of the flat fp32 gradient buffer back to back
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        for i in range(reps + 1):
            if i == 1:
                e0.record()
            tr.buckets.reset()
            for k in range(len(tr.buckets.ranges)):             # (the buckets' own launch path: fp32 slices or bf16 staging)
                tr.buckets._launch(k)
            tr.buckets.wait()
        e1.record(); e1.synchronize()
        ar_ms = e0.elapsed_time(e1) / reps
        # ... and what it costs inside the step: the same K steps with the exchange switched off (local gradients only).
        # NOTE: these steps apply LOCAL gradients, so the ranks' weights diverge -- this Trainer must not be used for anything
        # after this block (it is the last thing the pass does with it).
        k_off = min(steps, 20)
        saved, tr.buckets.active = tr.buckets.active, (lambda: False)
        fence()
        t0 = time.perf_counter()
        for _ in range(k_off):
            tr.train_step(x, lab, global_batch=B * world)
        fence()
        el_off = time.perf_counter() - t0
        tr.buckets.active = saved
        t = torch.tensor([el_off], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_off = float(t.item())
        nbytes = tr.gflat.numel() * (4 if args.grad_exchange == 'f32' else 2)
        exch = {'rccl_world': dist.get_world_size(), 'backend': dist.get_backend(), 'buckets': len(tr.buckets.ranges), 'bytes': nbytes,
                'dtype': args.grad_exchange,
                'allreduce_ms_per_step_standalone': round(ar_ms, 3),
                'allreduce_busbw_GBps': round(2.0 * (world - 1) / max(world, 1) * nbytes / (ar_ms * 1e-3) / 1e9, 1),
                'ms_per_step_without_exchange': round(el_off / k_off * 1e3, 4),
                'exposed_ms_per_step': round((el / steps - el_off / k_off) * 1e3, 4)}
    res = {
        'metric': 'training images/sec at %dx%d bs=%d per GPU (fwd + loss + bwd + train-mode BN + Adam)' % (size[0], size[1], B),
        'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'preheat_s': preheat_s, 'sustained_s': round(el, 3),
        'ms_per_step': round(el / steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[%d]: car/YOLO.py training step, Darknet-53 spec + 3-scale YOLO head '
                               '(A=3, C=30), synthetic render_car-style targets, %dx%d, bs=%d per GPU, %s activations / '
                               'fp32 master weights, BN statistics, loss and Adam' % (2 if world == 1 else 3, size[0], size[1], B, args.dtype),
                   'global_batch': B * world, 'image': list(size),
                   'parallelism': 'dp%d (batch-sharded; one RCCL all-reduce of the %d-element fp32 gradient bucket per step)'
                                  % (world, tr.gflat.numel()),
                   'gflop_per_image': round(fl / 1e9, 2)},
        'net_tflops': round(tf, 1), 'net_frac': round(tf / MFMA_PEAK_TFLOPS[args.dtype], 4),
        'final_losses': [round(float(v), 6) for v in losses.sum(dim=1).tolist()],
        'exchange': exch,
    }
    res.update(tinfo)
    if world == 1 and not args.no_roofline:
        # (round 6) roofline of the training step's dominant kernel FAMILY: BatchNorm backward = bn_reduce (reads dz, y) + bn_apply
        # (reads dz, y, writes dy) -- a third of the step's kernel time (profiles/*_train_kernel_stats.csv).  Algorithmic bytes =
        # 10 B per BatchNorm output element in bf16 (4 + 6; fp32: 20); durations = HIP events around every call, INSIDE the step
        # (beside the side stream's weight gradients), three steps after the timed region; `traffic` from a committed PMC pass
        # (profiles/*_train_pmc_traffic.json: bn_reduce + bn_apply<1,1> bytes per step) or null.
        tr.probe = []
        for _ in range(3):
            tr.train_step(x, lab)
        torch.cuda.synchronize()
        pr, tr.probe = tr.probe, None
        es = 2 if args.dtype == 'bf16' else 4
        nbytes = sum(p_[2] for p_ in pr) * 5 * es
        ms = sum(p_[3].elapsed_time(p_[4]) for p_ in pr)
        gbs = nbytes / (ms * 1e-3) / 1e9
        traffic, tsrc = None, None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_train_pmc_traffic.json')), reverse=True):
            with open(path) as f:
                prof = json.load(f)
            if prof.get('workload') == [B, size[0], size[1]]:
                ks = prof.get('kernels', {})
                tot = sum(v['hbm_bytes_per_launch'] * v.get('launches_per_step', 0) for k_, v in ks.items()
                          if k_.startswith('void bn_reduce_kernel') or k_.startswith('void bn_apply_kernel<bf16_t, 1') or k_.startswith('void bn_apply_kernel<float, 1'))
                if tot:
                    traffic, tsrc = int(tot), os.path.relpath(path, ROOT)
                    break
        res['roofline'] = {'kernel': 'BatchNorm backward (bn_reduce_kernel + bn_apply_kernel<.., 1, 1>), %d calls per step' % (len(pr) // 3),
                           'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
                           'algorithmic_bytes_per_step': nbytes // 3, 'ms_per_step': round(ms / 3, 3),
                           'traffic': traffic, 'traffic_unit': 'HBM bytes per step (FETCH_SIZE x2 + WRITE_SIZE)', 'traffic_source': tsrc,
                           'measured': 'HIP events around every yolo_bn_train_bwd_pp call of 3 steps, in the step (weight gradients on the side stream)'}
    res['plan'] = plan_report(args, tr.tuning_state())
    if dist is not None:
        res['per_rank_ms_per_step'] = [round(t_ / steps * 1e3, 4) for t_ in per_rank]
        res['tuning_identical_across_ranks'] = bool(tuning_same)
    return res


def plan_report(args, state):
    """What pinned the kernels of a pass: the plan file (and how many shapes it did not hold and were measured live), or the
    plan this box measured for itself (--tune measure)."""
    from yolo_amd import plans
    info = dict(getattr(args, 'plan_info', None) or {'file': None, 'mode': 'measure'})
    base = getattr(args, 'plan_state', None)
    info['measured_live'] = plans.new_keys(state, base) if base is not None else sum(len(state.get(s_, {})) for s_ in plans.SECTIONS)
    info['state_md5'] = plans.md5(state)          # (of the sections this pass uses: an inference pass holds no dgrad / wgrad choices)
    return info


def bench_train(args, spec, size, B, rank, world, dev, dist):
    out = train_pass(args, spec, size, B, rank, world, dev, dist, args.steps, args.warmup)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


def _free_port():
    import socket
    so = socket.socket()
    so.bind(('127.0.0.1', 0))
    port = so.getsockname()[1]
    so.close()
    return port


def relaunch_under_torchrun(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks on this node (one process per GPU) and hand the
    job over to them -- the command line the driver uses for N > 1."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    os.execvpe(cmd[0], cmd, env)


def launch_check(backend, rank, world, local):
    """--launch-check: everything of the N-rank path except the benchmark (rendezvous, barrier, MAX-reduce) -- the
    multi-process CPU test of the launcher (tests/test_dist_cpu.py, gloo)."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend)
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        ranks = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        if backend == 'nccl':
            dev = torch.device('cuda', local)
            t, ranks = t.to(dev), [r.to(dev) for r in ranks]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_gather(ranks, torch.tensor([rank], dtype=torch.int64, device=t.device))
        out = {'launch_check': True, 'n_gpus': world, 'world': dist.get_world_size(), 'max': float(t.item()),
               'ranks': [int(r.item()) for r in ranks], 'backend': backend}
        dist.barrier()
        dist.destroy_process_group()
    else:
        out = {'launch_check': True, 'n_gpus': 1, 'world': 1, 'max': 1.0, 'ranks': [0], 'backend': None}
    if rank == 0:
        emit(out)


def emit(out):
    """The ONE JSON line, as the LAST line of stdout: RCCL writes its version banner through C stdio, which sits in libc's
    buffer until exit when stdout is a pipe or a file -- flush it out first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def timed_pass(net, det, x, post, steps, warmup, fence, preheat_s=0.0, telemetry=None):
    """W untimed + K timed steps of forward + post-processing; returns (seconds for the K steps on this rank, telemetry dict).
    post: 'none' forward only | 'top1' decode + per-image arg-max + the (B, 6+C) rows copied to the host (the reference's
    predict owns its D2H, car/YOLO.py:597; pipelined two deep: no host stall) | 'top1_blocking' the same with a blocking
    .cpu() per step | 'nms' decode + per-class NMS (kept ids stay on the device).  preheat_s: run the same step for that
    many seconds first (clock / power state of a sustained run), outside the timed region."""
    pend = [None]

    def step(i=0):
        outs = net(x)
        if post == 'none':
            return outs
        if post == 'nms':
            if os.environ.get('YOLO_BENCH_NMS_TWO_CALLS'):          # (A/B knob: yolo_decode_scores + yolo_nms_from_scores)
                rows, scores = det.decode_scores(outs, mode='class')
                kept, ks, cnt = det.nms(rows, mode='class', scores=scores)
                return kept, cnt
            rows, scores, kept, ks, cnt = det.decode_nms(outs, mode='class')
            return kept, cnt
        if post == 'top1_blocking':
            return det.predict(outs)
        host, ev = det.predict_async(outs, slot=i & 1)
        if pend[0] is not None:
            pend[0].synchronize()                      # the rows of step i-1 are on the host before step i+1 re-uses their buffer
        pend[0] = ev
        return host

    if preheat_s > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < preheat_s:
            for i in range(10):
                step(i)
            torch.cuda.synchronize()
    for i in range(warmup):
        step(i)
    fence()
    if telemetry is not None:
        telemetry.start()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    fence()
    el = time.perf_counter() - t0
    return el, (telemetry.stop() if telemetry is not None else {})


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-child':      # (the cpu_baseline leg's pinned child process: see cpu_child)
        return cpu_child(json.loads(sys.argv[2]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (default 32; 64 in --mode train)')
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32', 'bf16x3', 'f16x3'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-fuse-stem', action='store_true', help='run the stem and the first down-sampling conv as two kernels (A/B)')
    ap.add_argument('--no-fuse-concat', action='store_true', help='up-sample + concat as a copy kernel instead of strided conv outputs (A/B)')
    ap.add_argument('--no-fuse-res', action='store_true', help='never use the fused residual-block kernel (A/B)')
    ap.add_argument('--no-fuse-tail', action='store_true', help='never fuse a 1x1 convolution into the 3x3 in front of it (A/B)')
    ap.add_argument('--no-side-stream', action='store_true', help='run the head tip/output convolutions on the main stream (A/B)')
    ap.add_argument('--tune-cache', default=None, help='JSON file remembering the measured per-layer kernel choices')
    ap.add_argument('--plan', default=None,
                    help='committed launch plan (yolo_amd/plans.py; default profiles/plan.json): the kernel variant of every layer shape, so '
                         'that this line, the rocprofv3 kernel trace and the PMC passes under profiles/ describe the same launches')
    ap.add_argument('--tune', default='plan', choices=['plan', 'measure'],
                    help="'plan' (default): launch the committed plan's kernels; a shape it does not hold is measured live and counted in "
                         "plan.measured_live.  'measure': ignore the plan, time every variant on this box (prints the plan it arrives at "
                         "as plan.md5; tools/make_plan.py writes one)")
    ap.add_argument('--post', default='top1', choices=['top1', 'nms', 'none'],
                    help="post-processing inside the timed step: 'top1' = the reference's predict (decode + per-image arg-max + D2H of"
                         " the rows); 'nms' = decode + per-class greedy NMS (BASELINE configs[4]); 'none' = the network forward alone")
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help="'infer' = BASELINE configs[1] (the headline metric); 'train' = configs[2]/[3]: one training "
                         "step (fwd + loss + bwd + train-mode BN + gradient all-reduce + Adam) per step")
    ap.add_argument('--no-northstar', action='store_true',
                    help='skip the extra 608x608 bs=64 passes (north-star shape, BASELINE configs[4] per-GPU shape) of the headline run')
    ap.add_argument('--no-train-key', action='store_true', help='skip the extra training-step pass (BASELINE configs[2]) of the headline run')
    ap.add_argument('--train-key', action='store_true', help='(default now) the training pass also runs under N > 1: BASELINE configs[3]')
    ap.add_argument('--no-repeats', action='store_true', help='skip the four extra K-step passes behind value_median')
    ap.add_argument('--no-f32-key', action='store_true', help='skip the extra fp32-path pass (the arithmetic the 1e-3 parity bar is tested on)')
    ap.add_argument('--sustain-steps', type=int, default=300,
                    help='timed steps of the SUSTAINED extra keys (608x608 bs 64; the training key runs 0.4x as many): each after a 2 s '
                         'pre-heat of the same pass and >= 20 warm-up steps.  0 = short legacy passes of --steps / 2 (tests)')
    ap.add_argument('--train-timeout', type=float, default=600.0,
                    help='watchdog (s) around the training pass under N > 1: when it fires the line is printed without the pass')
    ap.add_argument('--grad-exchange', default='f32', choices=['f32', 'bf16'],
                    help="N > 1 training: the dtype the gradient buckets travel in ('f32' = the reference's KVStore sum; 'bf16' = half the "
                         "bytes per xGMI link, yolo_amd/parallel.py:GradBuckets)")
    ap.add_argument('--grad-buckets', type=int, default=4, help='N > 1 training: number of buckets the 492 MB gradient buffer is cut into')
    ap.add_argument('--launch-check', action='store_true',
                    help='start the N ranks, rendezvous, barrier, MAX-reduce, print the world that ran, and exit (no benchmark)')
    args = ap.parse_args()

    backend = os.environ.get('YOLO_BENCH_BACKEND', 'nccl')          # ('gloo': the CPU test of the launcher path)
    need_gpu = not (args.launch_check and backend == 'gloo')
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if need_gpu and not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    # TEST-ONLY (tests/test_gpu_seam_dist.py): YOLO_BENCH_SHARED_GPU=1 with YOLO_BENCH_BACKEND=gloo lets the N ranks share the
    # GPUs that exist, so that the whole N > 1 path -- rendezvous, sharded passes, the Trainer's bucketed exchange between real
    # processes, the watchdog, the one JSON line -- runs on a one-GPU box.  The line is marked `shared_gpu_test`; it is not an
    # N-GPU measurement.
    shared = bool(os.environ.get('YOLO_BENCH_SHARED_GPU')) and backend == 'gloo'
    if need_gpu and torch.cuda.device_count() < args.gpus and not shared:
        raise SystemExit('--gpus %d but only %d GPU(s) are visible: refusing to report a %d-GPU number from fewer ranks'
                         % (args.gpus, torch.cuda.device_count(), args.gpus))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args.gpus, sys.argv[1:])               # does not return
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE %d != --gpus %d: n_gpus must be the number of ranks that ran' % (world, args.gpus))
    if args.launch_check:
        return launch_check(backend, rank, world, local)
    if shared:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1 or os.environ.get('YOLO_BENCH_FORCE_DIST'):      # (the env knob exercises the N>1 code path on one GPU)
        import torch.distributed as dist
        import datetime
        if shared:
            dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=max(60.0, args.train_timeout)))
        else:
            dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=max(60.0, args.train_timeout)))

    from yolo_amd.net import CarNet
    from yolo_amd.detect import Detector
    from yolo_amd.spec import darknet53_spec
    from yolo_amd import parallel

    from yolo_amd import plans
    spec = darknet53_spec()
    size = (args.size, args.size)
    B = args.batch or (64 if args.mode == 'train' else 32)
    plan_path = args.plan or plans.DEFAULT
    args.plan_state, args.plan_info = None, {'file': None, 'mode': args.tune}
    if args.tune == 'plan' and os.path.exists(plan_path):
        args.plan_state, meta = plans.load(plan_path)
        args.plan_info.update(file=os.path.relpath(plan_path, ROOT), md5=meta.get('md5'), commit=meta.get('commit'))
    if args.mode == 'train':
        return bench_train(args, spec, size, B, rank, world, dev, dist)
    net = CarNet(spec, dtype=args.dtype, device=dev, tune='measure', tune_cache=args.tune_cache,
                 fuse_stem=not args.no_fuse_stem, side_stream=not args.no_side_stream, fuse_concat=not args.no_fuse_concat,
                 fuse_res=not args.no_fuse_res, fuse_tail=not args.no_fuse_tail).initialize(seed=1234)
    if args.plan_state is not None:
        net.load_tuning_state(args.plan_state)
    net.prepare()
    det = Detector(spec, size, net.graph.steps(), device=dev)
    gen = torch.Generator(device='cpu').manual_seed(100 + rank)
    x = torch.rand((B, 3) + size, generator=gen).to(dev)            # synthetic images, resident in HBM
    errors = []

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def over_ranks(el):
        """(MAX over the ranks, [every rank's own seconds])"""
        if dist is None:
            return el, [el]
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = el
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per = [float(v) for v in t.tolist()]
        return max(per), per

    def pin_plan(n, Bp, sz):
        """N > 1: rank 0 measures the kernel variants of this shape, every rank adopts its choices (parallel.share_tuning) and
        the launch plans are compared -- all ranks run the same kernel instantiations.  -> (identical?, md5 of the plan)"""
        import hashlib
        if dist is not None and rank == 0:
            n.plan_signature(Bp, *sz)
        if dist is not None:
            parallel.share_tuning(n, src=0)
        sig = n.plan_signature(Bp, *sz)
        return parallel.same_on_all_ranks(sig), hashlib.md5(json.dumps(sig).encode()).hexdigest()[:12]

    plans_same, plan_md5 = pin_plan(net, B, size)
    # --warmup 0: one untimed step still runs first (as the training pass does): the first forward of a shape builds its launch
    # plan and MEASURES the kernel variants -- set-up, not a step; reported as `setup_steps`
    setup_steps = 1 if args.warmup == 0 else 0
    # `value` times the reference's LITERAL predict (car/YOLO.py:568-597): the rows are copied to the host by a blocking copy in
    # every step, as rounds 1-3 measured it; the pipelined copy (pinned buffers, two deep) is `value_async_predict`
    post_main = 'top1_blocking' if args.post == 'top1' else args.post
    el, per_rank = over_ranks(timed_pass(net, det, x, post_main, args.steps, args.warmup + setup_steps, fence)[0])
    ms_per_step = el / args.steps * 1e3
    value = world * B * args.steps / el
    # the same K-step pass four more times: `value` stays the first pass (what the driver's clock brackets), the median of
    # the five tells a 1-3 % change from run-to-run noise
    reps = [value] + [world * B * args.steps / over_ranks(timed_pass(net, det, x, post_main, args.steps, 0, fence)[0])[0]
                      for _ in range(0 if args.no_repeats else 4)]
    post_name = {'nms': 'per-class NMS', 'top1': 'top-1 + blocking D2H of the rows', 'none': 'nothing (forward alone)'}[args.post]

    out = {
        'metric': 'images/sec at %dx%d bs=%d per GPU (Darknet-53 spec + 3-scale YOLO head forward, anchor '
                  'decode + %s)' % (size[0], size[1], B, post_name),
        'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic', 'setup_steps': setup_steps,
        'config': {'workload': 'BASELINE configs[1]: Darknet-53 spec layers [1,2,8,8,4] channels [32..1024] + '
                               '3-scale YOLO head (A=3, C=30) forward, random Xavier weights, %dx%d, bs=%d per GPU, '
                               '+ decode/%s' % (size[0], size[1], B, 'per-class NMS (valid 0.01, IoU 0.45, top-k 400, keep 100)' if args.post == 'nms' else post_name),
                   'global_batch': B * world, 'image': list(size),
                   'parallelism': 'dp%d (batch-sharded, no data-path collective)' % world,
                   'gflop_per_image': round(net.graph.flops(*size) / 1e9, 2),
                   # what `value` is NOT: inside the north-star 1e-3 unless the dtype says so (tests/test_gpu_boxes.py)
                   'arithmetic': {'bf16': 'bf16 MFMA, fp32 accumulate: decoded boxes ~1e-2 (RMS 7e-3) of the fp32 oracle, top-1 index may differ; '
                                          'the number INSIDE the 1e-3 tolerance is `parity_path` (dtype bf16x3)',
                                  'f16': "IEEE half (the reference's use_fp16): boxes 1.6e-2 max / 6e-4 RMS of the fp32 oracle",
                                  'bf16x3': 'split bf16, three bf16 MFMAs per product: decoded boxes <= 1e-3 of the fp32 oracle (3e-4 observed), top-1 indices identical',
                                  'f16x3': 'split f16, three f16 MFMAs per product: decoded boxes <= 1e-3 of the fp32 oracle (6e-5 observed: the fp32 path\'s own), top-1 indices identical',
                                  'f32': 'exact-fp32 MFMA: decoded boxes <= 1e-3 of the fp32 oracle (7e-5 observed), indices bit-exact'}[args.dtype]},
    }
    out['value_median'] = round(float(np.median(reps)), 2)
    out['value_repeats'] = [round(v, 1) for v in reps]
    if args.post == 'top1' and not args.no_repeats:
        # the same pass with the rows copied asynchronously into pinned memory, two buffers deep (Detector.predict_async): the
        # copy of step i overlaps the launches of step i + 1 -- a host-side gain, not kernel work, hence not `value`
        ela = over_ranks(timed_pass(net, det, x, 'top1', args.steps, 0, fence)[0])[0]
        out['value_async_predict'] = round(world * B * args.steps / ela, 2)
    # md5 over [(op, kernel instantiation)] of the launch plan that ran: what a kernel trace / PMC pass must carry to describe THIS run
    out['plan_md5'] = plan_md5
    if dist is not None:
        # stragglers and rank-dependent plans, visible the day a node exists: every rank's own time for the K steps and
        # whether all ranks launch the same kernel instantiations (rank 0 measured, the others adopted its choices)
        out['per_rank_ms_per_step'] = [round(t / args.steps * 1e3, 4) for t in per_rank]
        out['rank_ms_per_step_min_max'] = [round(min(per_rank) / args.steps * 1e3, 4), round(max(per_rank) / args.steps * 1e3, 4)]
        out['plans_identical_across_ranks'] = bool(plans_same)
    if shared:
        out['shared_gpu_test'] = 'TEST ONLY: %d ranks share %d GPU(s) over gloo -- not an N-GPU measurement' % (world, torch.cuda.device_count())
    out['plan'] = plan_report(args, net.tuning_state())
    out['plan']['stale'] = net.stale_choices     # (plan entries this library build refused: dropped, measured live)
    out['net_tflops'] = round(net.graph.flops(*size) * value / 1e12 / world, 1)          # per GPU
    out['net_frac'] = round(out['net_tflops'] / MFMA_PEAK_TFLOPS[args.dtype], 4)           # whole pass vs the dense MFMA peak
    out['net_frac_median'] = round(net.graph.flops(*size) * out['value_median'] / 1e12 / world / MFMA_PEAK_TFLOPS[args.dtype], 4)

    if rank == 0 and not args.no_roofline:
        kernels = net.plan_kernels(B, *size)
        info = {n: (k, f) for n, k, f in kernels}
        hist = {}
        for n, k, f in kernels:
            hist[k] = hist.get(k, 0) + 1
        agg = {}
        for _ in range(args.steps):
            ev = []
            net.forward_timed(x, ev)
            torch.cuda.synchronize()
            for name, e0, e1 in ev:
                k, f = info[name]
                a = agg.setdefault(k, [0.0, 0, 0])
                a[0] += e0.elapsed_time(e1) * 1e-3
                a[1] += 1
                a[2] += f
        dom = max((k for k in agg if agg[k][2] > 0), key=lambda k: agg[k][0])
        tsec, nl, fl = agg[dom]
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        ach = fl / tsec / 1e12
        traffic, traffic_src, traffic_note = pmc_traffic(dom, B, size, plan_md5, nl // args.steps)
        nbytes = net.plan_bytes(B, *size)
        alg_bytes = sum(nbytes.get(n, 0) for n, (k, f) in info.items() if k == dom) // max(nl // args.steps, 1)
        out['roofline'] = {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s',
                           'frac': round(ach / peak, 4), 'traffic': traffic,
                           # (a committed rocprofv3 --pmc pass of THIS launch plan, not a counter read of this run)
                           'traffic_source': traffic_src,
                           # input + output (+ residual) + weights once, mean over this kernel's launches: what `traffic` would be
                           # with no re-fetch
                           'algorithmic_bytes': int(alg_bytes), 'kernel': dom,
                           'launches_per_step': nl // args.steps,
                           'avg_launch_us': round(tsec / nl * 1e6, 2),
                           'flops_per_launch': fl // nl}
        if traffic_note:
            out['roofline']['traffic_note'] = traffic_note
        out['kernels'] = [{'kernel': k, 'launches_per_step': v[1] // args.steps,
                           'ms_per_step': round(v[0] / args.steps * 1e3, 4),
                           'tflops': round(v[2] / v[0] / 1e12, 1) if v[2] else None}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])]
    headline = (args.size, B, args.post) == (416, 32, 'top1') and args.dtype == 'bf16'
    sustain = args.sustain_steps > 0
    if not args.no_northstar and headline:
        # The north-star shape (BASELINE.json: ">= 40 % of bf16 MFMA peak on the Darknet-53 forward at 608x608 bs=64") and
        # BASELINE configs[4]'s per-GPU shape (the same + decode + per-class NMS), timed in this same process so that the
        # driver-run line carries them: same net object, new launch plan.  SUSTAINED: 2 s of the same pass first, >= 20 warm-up
        # steps, >= 300 timed steps (yolo_gluon.py:317-331 times 10 + 100 iterations), clock and power sampled alongside.
        del x
        size6, B6 = (608, 608), 64
        det6 = Detector(spec, size6, net.graph.steps(), device=dev)
        x6 = torch.rand((B6, 3) + size6, generator=gen).to(dev)
        if sustain:
            k6, w6, heat = max(args.steps, args.sustain_steps), max(args.warmup, 20), 2.0
        else:
            k6, w6, heat = max(args.steps // 2, 3), max(args.warmup // 2, 2), 0.0
        fl6 = net.graph.flops(*size6)
        same6, md5_6 = pin_plan(net, B6, size6)
        # (`northstar_608_forward`: the network forward alone -- the quantity BASELINE.json's target is worded on, "the Darknet-53
        #  forward at 608x608 bs=64"; the other two keys add the post-processing to the timed step)
        for key, post in (('northstar_608_forward', 'none'), ('northstar_608', 'top1'), ('northstar_608_nms', 'nms')):
            tel = Telemetry(local) if (rank == 0 and sustain) else None
            el_, tinfo = timed_pass(net, det6, x6, post, k6, w6, fence, preheat_s=heat, telemetry=tel)
            el6, per6 = over_ranks(el_)
            v6 = world * B6 * k6 / el6
            tf6 = fl6 * v6 / 1e12 / world
            out[key] = {'workload': 'D53 spec forward 608x608 bs=64 per GPU' + ('' if post == 'none' else ' + decode/%s' % ('per-class NMS' if post == 'nms' else 'top-1 + D2H')),
                        'value': round(v6, 2), 'unit': 'images/s', 'steps': k6, 'warmup': w6, 'preheat_s': heat,
                        'sustained_s': round(el6, 3), 'ms_per_step': round(el6 / k6 * 1e3, 4), 'net_tflops': round(tf6, 1),
                        'frac_of_peak': round(tf6 / MFMA_PEAK_TFLOPS[args.dtype], 4), 'gflop_per_image': round(fl6 / 1e9, 2)}
            out[key].update(tinfo)
            if dist is not None:
                out[key]['per_rank_ms_per_step'] = [round(t / k6 * 1e3, 4) for t in per6]
                out[key]['plans_identical_across_ranks'] = bool(same6)
        del x6
    if not args.no_f32_key and headline and world == 1:
        # what the arithmetic of the 1e-3 parity bar costs: the SAME workload on the fp32 path (exact-f32 MFMA, fp32
        # activations; tests/test_gpu_configs.py holds it to <= 1e-3 of the oracle)
        del net
        torch.cuda.empty_cache()
        net32 = CarNet(spec, dtype='f32', device=dev, tune='measure').initialize(seed=1234)
        net32.prepare()
        x32 = torch.rand((B, 3) + size, generator=gen).to(dev)
        k32 = max(args.steps // 2, 3)
        el32 = timed_pass(net32, det, x32, 'top1', k32, 2, fence)[0]
        v32 = B * k32 / el32
        out['f32_path'] = {'workload': 'the headline workload on the fp32 parity path (<= 1e-3 vs the oracle)', 'value': round(v32, 2),
                           'unit': 'images/s', 'steps': k32, 'ms_per_step': round(el32 / k32 * 1e3, 4),
                           'net_tflops': round(net32.graph.flops(*size) * v32 / 1e12, 1),
                           'frac_of_f32_peak': round(net32.graph.flops(*size) * v32 / 1e12 / MFMA_PEAK_TFLOPS['f32'], 4)}
        # (round 6) "the number inside the tolerance": the same workload on the SPLIT bf16 path (dtype 'bf16x3': (hi, lo) bf16 pairs,
        # three bf16 MFMAs per product) -- decoded boxes within 1e-3 of the fp32 oracle (tests/test_gpu_boxes.py, cpu_baseline.box_parity)
        del net32
        torch.cuda.empty_cache()
        netx3 = CarNet(spec, dtype='bf16x3', device=dev, tune='measure').initialize(seed=1234)
        if args.plan_state is not None:
            netx3.load_tuning_state(args.plan_state)
        netx3.prepare()
        kx3 = max(args.steps, 10)
        elx3 = min(timed_pass(netx3, det, x32, 'top1_blocking', kx3, 3, fence)[0] for _ in range(2))
        vx3 = B * kx3 / elx3
        flx3 = netx3.graph.flops(*size)
        out['parity_path'] = {'workload': "the headline workload with dtype='bf16x3' (split bf16: decoded boxes <= 1e-3 vs the fp32 oracle)",
                              'value': round(vx3, 2), 'unit': 'images/s', 'steps': kx3, 'ms_per_step': round(elx3 / kx3 * 1e3, 4),
                              'net_tflops': round(flx3 * vx3 / 1e12, 1),
                              'mfma_tflops': round(3 * flx3 * vx3 / 1e12, 1),           # (what the matrix pipe executes: 3 MFMAs per product)
                              'frac_of_bf16_peak_executed': round(3 * flx3 * vx3 / 1e12 / MFMA_PEAK_TFLOPS['bf16'], 4),
                              'vs_f32_path': round(vx3 / v32, 2),
                              'plan': plan_report(args, netx3.tuning_state()) if args.plan_state is not None else None}
        if world == 1 and not args.no_northstar:
            # ... and BASELINE configs[4]'s per-GPU shape on the same path: 608x608 bs 64 + decode + per-class NMS, inside the tolerance
            size6, B6 = (608, 608), 64
            det6x = Detector(spec, size6, netx3.graph.steps(), device=dev)
            x6x = torch.rand((B6, 3) + size6, generator=gen).to(dev)
            k6x = max(args.steps // 2, 5)
            el6x = timed_pass(netx3, det6x, x6x, 'nms', k6x, 2, fence)[0]
            v6x = B6 * k6x / el6x
            fl6x = netx3.graph.flops(*size6)
            out['parity_path']['northstar_608_nms'] = {'workload': 'D53 spec forward 608x608 bs=64 + decode/per-class NMS, dtype bf16x3',
                                                       'value': round(v6x, 2), 'unit': 'images/s', 'steps': k6x, 'ms_per_step': round(el6x / k6x * 1e3, 4),
                                                       'net_tflops': round(fl6x * v6x / 1e12, 1),
                                                       'frac_of_bf16_peak_executed': round(3 * fl6x * v6x / 1e12 / MFMA_PEAK_TFLOPS['bf16'], 4)}
            del x6x, det6x
        # ... and the IEEE-half sibling of the split path (dtype 'f16x3': boxes at the fp32 path's own distance from the oracle)
        del netx3
        torch.cuda.empty_cache()
        netx3 = CarNet(spec, dtype='f16x3', device=dev, tune='measure').initialize(seed=1234)
        if args.plan_state is not None:
            netx3.load_tuning_state(args.plan_state)
        netx3.prepare()
        elh = min(timed_pass(netx3, det, x32, 'top1_blocking', kx3, 3, fence)[0] for _ in range(2))
        out['parity_path']['f16x3'] = {'value': round(B * kx3 / elh, 2), 'unit': 'images/s', 'ms_per_step': round(elh / kx3 * 1e3, 4),
                                       'note': 'split f16 (22 significant bits): decoded boxes ~9e-5 of the fp32 oracle; needs activations inside IEEE half range'}
        net32 = netx3
        # ... and the reference's own reduced precision (use_fp16, car/YOLO.py:98-100) on the same workload: the bf16 MFMA rate minus
        # what the power cap takes, 12x closer to the fp32 oracle on the decoded boxes (cpu_baseline.box_parity)
        del net32
        torch.cuda.empty_cache()
        net16 = CarNet(spec, dtype='f16', device=dev, tune='measure').initialize(seed=1234)
        if args.plan_state is not None:
            net16.load_tuning_state(args.plan_state)
        net16.prepare()
        k16 = max(args.steps, 10)
        # (two passes, the faster one: one refresh of round 5 saw this pass alone at 7.5 ms per step between two runs at 4.2)
        el16s = [timed_pass(net16, det, x32, 'top1_blocking', k16, 3, fence)[0] for _ in range(2)]
        el16 = min(el16s)
        v16 = B * k16 / el16
        out['f16_path'] = {'workload': "the headline workload with dtype='f16' (the reference's use_fp16)", 'value': round(v16, 2), 'unit': 'images/s',
                           'steps': k16, 'ms_per_step': round(el16 / k16 * 1e3, 4), 'passes_ms_per_step': [round(e / k16 * 1e3, 4) for e in el16s],
                           'net_tflops': round(net16.graph.flops(*size) * v16 / 1e12, 1),
                           'frac_of_peak': round(net16.graph.flops(*size) * v16 / 1e12 / MFMA_PEAK_TFLOPS['f16'], 4)}
        net = net16
        del x32
    if not args.no_train_key and headline:
        # BASELINE configs[2] (training step, 416x416 bs=64 per GPU) in the same driver-run line; under N > 1 this is
        # configs[3]: the RCCL all-reduce of the gradient buckets inside the step.  A collective that hangs must not cost
        # the headline line: a watchdog on every rank prints the line without the pass and ends the process -- with a
        # NON-ZERO status and a top-level `errors` entry, so that a failed configs[3] run cannot pass for a clean one.
        del net
        torch.cuda.empty_cache()
        import threading

        def give_up():
            if rank == 0:
                msg = 'the training pass did not finish within %.0f s (watchdog)' % args.train_timeout
                out['train_416_bs64'] = {'error': msg, 'n_gpus': world}
                out['errors'] = errors + ['train_416_bs64: ' + msg]
                emit(out)
            os._exit(3)

        dog = threading.Timer(args.train_timeout, give_up) if world > 1 else None
        if dog is not None:
            dog.daemon = True
            dog.start()
        try:
            if sustain:
                kt, wt, heat = max(args.steps, int(0.4 * args.sustain_steps)), max(args.warmup, 20), 2.0
            else:
                kt, wt, heat = max(args.steps // 2, 5), 2, 0.0
            # N = 1: BASELINE configs[2], 64 images on the GPU.  N > 1: configs[3] = the SAME training loop at GLOBAL batch 256
            # sharded over the ranks (car/YOLO.py:372-396, yolo_gluon.py:100-124: 8 x 32) -- 256 // N images per GPU, strong
            # scaling of that job; the 64-per-GPU pass (weak scaling of configs[2]) is the extra key `train_416_bs64_weak`.
            Bt = 64 if world == 1 else max(256 // world, 1)
            keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'preheat_s', 'sustained_s', 'ms_per_step', 'net_tflops',
                    'net_frac', 'final_losses', 'exchange', 'sclk_mhz', 'power_w', 'telemetry_samples', 'telemetry_source',
                    'per_rank_ms_per_step', 'tuning_identical_across_ranks', 'plan', 'roofline')
            t = train_pass(args, spec, size, Bt, rank, world, dev, dist, kt, wt, preheat_s=heat,
                           telemetry=Telemetry(local) if (rank == 0 and sustain) else None)
            out['train_416_bs64'] = {k: t[k] for k in keep if k in t}
            out['train_416_bs64']['workload'] = t['config']['workload']
            out['train_416_bs64']['global_batch'] = t['config']['global_batch']
            out['train_416_bs64']['batch_per_gpu'] = Bt
            if world > 1 and Bt != 64:
                import gc
                del t
                gc.collect()                              # (net <-> trainer reference cycle: the first pass's 10-40 GB must go first)
                torch.cuda.empty_cache()
                t = train_pass(args, spec, size, 64, rank, world, dev, dist, max(kt // 2, 3), wt, preheat_s=0.0)
                out['train_416_bs64_weak'] = {k: t[k] for k in keep if k in t}
                out['train_416_bs64_weak']['workload'] = t['config']['workload'] + ' (64 per GPU: weak scaling of configs[2])'
                out['train_416_bs64_weak']['global_batch'] = t['config']['global_batch']
                out['train_416_bs64_weak']['batch_per_gpu'] = 64
        except Exception as e:                                   # (a rank-local failure: the other ranks meet the watchdog)
            msg = '%s: %s' % (type(e).__name__, e)
            out['train_416_bs64'] = {'error': msg, 'n_gpus': world}
            errors.append('train_416_bs64: ' + msg)
            if world > 1:
                if rank == 0:
                    out['errors'] = errors
                    emit(out)
                os._exit(3)
        finally:
            if dog is not None:
                dog.cancel()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(size, dev=dev, plan_state=args.plan_state, plan_batch=B)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if errors:
        out['errors'] = errors
    if rank == 0:
        emit(out)
    if errors:
        sys.exit(3)


if __name__ == '__main__':
    main()
