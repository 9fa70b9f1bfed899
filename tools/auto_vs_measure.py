#!/usr/bin/env python
"""Forward time with the library's built-in variant heuristic (tune='auto') vs measured selection."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 416
dev = torch.device('cuda:0')
x = torch.rand((B, 3, S, S), device=dev)
for tune in ('auto', 'measure', 'auto', 'measure'):
    net = CarNet(darknet53_spec(), dtype='bf16', device=dev, tune=tune).initialize(1); net.prepare()
    for _ in range(5): net(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): net(x)
    torch.cuda.synchronize()
    print('%-8s %.3f ms/step' % (tune, (time.perf_counter() - t0) / 30 * 1e3))
    if tune == 'measure' and '--dump' in sys.argv:
        for (kind, d, name) in net._plans[(B, S, S)].ops:
            if kind == 'conv': print('   ', name, d.Cin, d.Cout, d.ksize, d.stride, 'algo', d.algo)
