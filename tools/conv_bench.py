#!/usr/bin/env python
"""Time every conv algo on every distinct layer shape of the D53 net (HIP events, L2-cold-ish:
each shape's buffers are private).  Prints TFLOP/s per (layer shape, algo)."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--size', type=int, default=416)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--algos', default='1,2,3,4,5,6')
ap.add_argument('--only', default='', help='comma list of layer names')
a = ap.parse_args()
dev = torch.device('cuda:0')
net = CarNet(darknet53_spec(), dtype=a.dtype, device=dev).initialize(1)
x = torch.rand((a.batch, 3, a.size, a.size), device=dev)
net(x)
lib = L.load()
plan = net._plans[(a.batch, a.size, a.size)]
kern = net.plan_kernels(a.batch, a.size, a.size)
seen = {}
algos = [int(s) for s in a.algos.split(',')]
print('%-18s %5s %5s %2s %2s | %s' % ('in', 'Cin', 'Cout', 'k', 's', '  '.join('a%d:TF(us)' % g for g in algos)))
st = torch.cuda.current_stream().cuda_stream
for (kind, d, name), (_, kname, fl) in zip(plan.ops, kern):
    if kind != 'conv':
        continue
    if a.only and name not in a.only.split(','):
        continue
    key = (d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.out_f32, bool(d.residual))
    if key in seen:
        continue
    seen[key] = 1
    res = []
    for g in algos:
        d.algo = g
        rc = lib.yolo_conv_fwd(C.byref(d), st)
        if rc != 0:
            res.append('   --    ')
            continue
        for _ in range(3):
            lib.yolo_conv_fwd(C.byref(d), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            lib.yolo_conv_fwd(C.byref(d), st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        res.append('%5.0f(%4.0f)' % (fl / us / 1e6, us))
    d.algo = 0
    print('%-18s %5d %5d %2d %2d | %s   %s' % ('%dx%dx%d' % (d.N, d.H, d.W), d.Cin, d.Cout, d.ksize, d.stride, '  '.join(res), name))
