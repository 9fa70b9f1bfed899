#!/usr/bin/env python
"""Does splitting the batch over two streams fill the idle time of the latency-bound layers?  One CarNet at bs B against two CarNets
at bs B/2 running concurrently on two streams (same weights), forward only.    python tools/two_stream_infer.py [--size 416 --batch 32]"""
import argparse, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32); ap.add_argument('--size', type=int, default=416)
a = ap.parse_args()
dev = torch.device('cuda:0')
def timed(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
one = CarNet(darknet53_spec(), dtype='bf16', device=dev, tune='measure').initialize(1)
x = torch.rand((a.batch, 3, a.size, a.size), device=dev)
one(x)
t1 = min(timed(lambda: one(x)), timed(lambda: one(x)))
print('one stream,  bs %d: %.3f ms/step = %.1f img/s' % (a.batch, t1, a.batch / t1 * 1e3))
h = a.batch // 2
nets = [CarNet(darknet53_spec(), dtype='bf16', device=dev, tune='measure').initialize(1) for _ in range(2)]
xs = [x[:h].contiguous(), x[h:].contiguous()]
for n_, x_ in zip(nets, xs): n_(x_)
th = min(timed(lambda: nets[0](xs[0])), timed(lambda: nets[0](xs[0])))
print('one stream,  bs %d: %.3f ms/step = %.1f img/s' % (h, th, h / th * 1e3))
ss = [torch.cuda.Stream(device=dev) for _ in range(2)]
def both():
    main = torch.cuda.current_stream()
    for s in ss: s.wait_stream(main)
    for n_, x_, s in zip(nets, xs, ss):
        with torch.cuda.stream(s): n_(x_)
    for s in ss: main.wait_stream(s)
t2 = min(timed(both), timed(both))
print('two streams, 2 x bs %d: %.3f ms/step = %.1f img/s' % (h, t2, a.batch / t2 * 1e3))
