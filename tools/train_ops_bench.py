#!/usr/bin/env python
"""Time the training-only kernels (train-mode BN forward/backward, weight gradient) on the layer shapes of
the D53 spec at 416x416 (BASELINE config 3 shape family) and print GB/s or TFLOP/s per shape."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--what', default='bn,wgrad')
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--algos', default='0', help='weight-gradient kernels to time (yolo_conv_wgrad_algo ids)')
ap.add_argument('--k3s1', action='store_true', help='weight gradient: only the 3x3 stride-1 shapes with Cin >= 64')
ap.add_argument('--k1', action='store_true', help='weight gradient: only the 1x1 shapes')
ap.add_argument('--s2', action='store_true', help='weight gradient: only the stride-2 shapes')
ap.add_argument('--cold', action='store_true', help='evict L2 / MALL (1 GiB fill) before every timed launch')
a = ap.parse_args()
lib = L.load()
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr() if t is not None else None


def timed(fn):
    for _ in range(3):
        fn()
    if a.cold:
        global _scratch
        if '_scratch' not in globals():
            _scratch = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        tot = 0.0
        for _ in range(a.iters):
            _scratch.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot * 1e3 / a.iters
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.iters          # us


# (H=W of the conv output, C) of the BN'd tensors, D53 416x416
BN_SHAPES = [(416, 32), (208, 64), (208, 32), (104, 128), (104, 64), (52, 256), (52, 128), (26, 512), (26, 256),
             (13, 1024), (13, 512)]
if 'bn' in a.what:
    for hw, C in BN_SHAPES:
        npix = a.batch * hw * hw
        y = torch.randn((npix, C), device=dev).bfloat16()
        dz = torch.randn((npix, C), device=dev).bfloat16()
        z = torch.empty_like(y)
        g, b = torch.rand(C, device=dev) + .5, torch.randn(C, device=dev)
        mean, inv, rm, rv = (torch.zeros(C, device=dev) for _ in range(4))
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = torch.zeros(24 * C,
                         dtype=torch.uint8, device=dev)
        tf = timed(lambda: L.check(lib.yolo_bn_train_fwd(p(y), p(g), p(b), None, p(z), p(mean), p(inv), p(rm), p(rv),
                                                          p(ws), npix, C, 1e-5, 0.9, 0.1, 1, st), "bn_fwd"))
        tb = timed(lambda: L.check(lib.yolo_bn_train_bwd(p(dz), p(y), p(mean), p(inv), p(g), p(b), p(z), p(dg), p(db),
                                                          p(ws), npix, C, 0.1, 1, st), "bn_bwd"))
        by = npix * C * 2
        print('bn %4d^2 C=%4d  fwd %7.1f us (%5.2f TB/s of 3 passes)  bwd %7.1f us (%5.2f TB/s of 5 passes)' %
              (hw, C, tf, 3 * by / tf / 1e6, tb, 5 * by / tb / 1e6), flush=True)

# (Ho=Wo, Cin, Cout, k, stride)
WG_SHAPES = [(104, 64, 128, 3, 2), (208, 32, 64, 3, 2), (208, 32, 64, 3, 1), (208, 64, 32, 1, 1), (104, 64, 128, 3, 1), (104, 128, 64, 1, 1),
             (52, 128, 256, 3, 1), (52, 256, 128, 1, 1), (26, 256, 512, 3, 1), (26, 512, 256, 1, 1),
             (13, 512, 1024, 3, 1), (13, 1024, 512, 1, 1), (416, 8, 32, 3, 1),
             # head layers, and the 608x608 family's widths
             (13, 1024, 2048, 3, 1), (26, 512, 1024, 3, 1), (52, 256, 512, 3, 1), (19, 512, 1024, 3, 1), (38, 256, 512, 3, 1),
             (76, 128, 256, 3, 1), (152, 64, 128, 3, 1),
             # 1x1 head layers
             (13, 2048, 1024, 1, 1), (26, 1024, 512, 1, 1), (52, 512, 256, 1, 1),
             # the deeper stride-2 layers
             (52, 128, 256, 3, 2), (26, 256, 512, 3, 2), (13, 512, 1024, 3, 2)]
if 'wgrad' in a.what:
    for ho, ci, co, k, s in WG_SHAPES:
        if a.k1 and k != 1:
            continue
        if a.k3s1 and not (k == 3 and s == 1 and ci >= 64):
            continue
        if a.s2 and s != 2:
            continue
        H = ho * s
        x = torch.randn((a.batch, H, H, ci), device=dev).bfloat16()
        dy = torch.randn((a.batch, ho, ho, co), device=dev).bfloat16()
        dw = torch.zeros((co, ci, k, k), device=dev)
        ws = torch.zeros(max(lib.yolo_conv_wgrad_workspace_bytes(co, ci, k, 1), 16), dtype=torch.uint8, device=dev)
        fl = 2.0 * a.batch * ho * ho * ci * co * k * k
        for algo in [int(v) for v in a.algos.split(',')]:
            rc = lib.yolo_conv_wgrad_algo(p(dy), p(x), p(dw), a.batch, H, H, ci, co, k, s, co, 1, p(ws), algo, st)
            if rc == L.EUNSUPPORTED:
                continue
            L.check(rc, 'wgrad')
            t = timed(lambda: L.check(lib.yolo_conv_wgrad_algo(p(dy), p(x), p(dw), a.batch, H, H, ci, co, k, s, co, 1, p(ws), algo, st), "wgrad"))
            print('wgrad %3d^2 %4d->%4d k%d s%d algo %d  %7.1f us  %6.1f TFLOP/s' % (ho, ci, co, k, s, algo, t, fl / t / 1e6), flush=True)
