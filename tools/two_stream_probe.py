#!/usr/bin/env python
"""Does running the batch as two half-batches on two HIP streams (kernels of the halves overlap and fill each
other's tails) beat one full-batch stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 416
dev = torch.device('cuda:0')
spec = darknet53_spec()
full = CarNet(spec, dtype='bf16', device=dev, tune='measure').initialize(1); full.prepare()
h1 = CarNet(spec, dtype='bf16', device=dev, tune='measure').initialize(1); h1.prepare()
h2 = CarNet(spec, dtype='bf16', device=dev, tune='measure').initialize(1); h2.prepare()
x = torch.rand((B, 3, S, S), device=dev)
xa, xb = x[:B // 2].contiguous(), x[B // 2:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_full():
    full(x)
def run_halves():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        h1(xa)
    with torch.cuda.stream(s2):
        h2(xb)
    cur.wait_stream(s1); cur.wait_stream(s2)
for f, name in ((run_full, 'one stream, full batch'), (run_halves, 'two streams, half batches')):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): f()
    torch.cuda.synchronize()
    print('%-28s %.3f ms/step' % (name, (time.perf_counter() - t0) / 30 * 1e3))
