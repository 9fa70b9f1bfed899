#!/usr/bin/env python
"""Per-layer timing of the forward (HIP events around every launch): ms, TFLOP/s, and the
HBM-roofline time (input + output + weights once, at 6.3 TB/s achievable) for each conv."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--size', type=int, default=416)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--tune', default='measure')
a = ap.parse_args()
dev = torch.device('cuda:0')
net = CarNet(darknet53_spec(), dtype=a.dtype, device=dev, tune=a.tune).initialize(1)
x = torch.rand((a.batch, 3, a.size, a.size), device=dev)
for _ in range(3):
    net(x)
kern = net.plan_kernels(a.batch, a.size, a.size)
plan = net._plans[(a.batch, a.size, a.size)]
es = 2 if a.dtype in ('bf16', 'f16') else 4          # (bf16x3: two 2-byte planes)
tot = {}
for _ in range(a.iters):
    ev = []
    net.forward_timed(x, ev)
    torch.cuda.synchronize()
    for name, e0, e1 in ev:
        tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
print('%-22s %-14s %5s %5s %4s %4s %8s %8s %8s %6s' % ('layer', 'in(NHWC)', 'Cin', 'Cout', 'k', 's', 'us', 'TF', 'hbm_us', 'bound'))
sum_us = sum_roof = 0
for (kind, payload, name), (_, kname, fl) in zip(plan.ops, kern):
    us = tot[name] / a.iters * 1e3
    if kind != 'conv':
        print('%-22s %58s %8.1f' % (name, kind, us)); sum_us += us; continue
    d = payload
    pad = d.ksize // 2
    ho, wo = (d.H + 2 * pad - d.ksize) // d.stride + 1, (d.W + 2 * pad - d.ksize) // d.stride + 1
    byt = d.N * d.H * d.W * d.Cin * es + d.N * ho * wo * d.Cout * (4 if d.out_f32 else es) + d.Cout * d.Cin * d.ksize ** 2 * es
    if d.residual: byt += d.N * ho * wo * d.Cout * es
    hbm_us = byt / 6.3e12 * 1e6
    mf_us = fl / {'bf16': 2.5e15, 'f16': 2.5e15, 'bf16x3': 2.5e15 / 3, 'f16x3': 2.5e15 / 3}.get(a.dtype, 157e12) * 1e6
    roof = max(hbm_us, mf_us)
    sum_us += us; sum_roof += roof
    print('%-22s %-14s %5d %5d %4d %4d %8.1f %8.1f %8.1f %6s %5.0f%% a%d' % (name, '%dx%dx%d' % (d.N, d.H, d.W), d.Cin, d.Cout, d.ksize, d.stride,
          us, fl / us / 1e6, hbm_us, 'hbm' if hbm_us > mf_us else 'mfma', 100 * roof / us, d.algo))
print('total %.1f us; sum of per-layer rooflines %.1f us' % (sum_us, sum_roof))
