#!/usr/bin/env python
"""Time one training step (BASELINE config 3 shape family): D53 spec, fwd + loss + bwd + Adam."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolo_amd.net import CarNet
from yolo_amd.train import Trainer
from yolo_amd.spec import darknet53_spec
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--size', type=int, default=416)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--tune', default='measure')
a = ap.parse_args()
dev = torch.device('cuda:0')
net = CarNet(darknet53_spec(), dtype=a.dtype, device=dev, tune=a.tune).initialize(1)
tr = Trainer(net, (a.size, a.size))
x = torch.rand((a.batch, 3, a.size, a.size), device=dev)
rng = np.random.default_rng(3)
lab = -np.ones((a.batch, 1, 30), np.float32)
for b in range(a.batch):
    if rng.random() < 0.5:
        continue
    d = rng.random(24).astype(np.float32); d /= d.sum()
    lab[b, 0, :6] = [int(np.argmax(d)), rng.uniform(.15, .85), rng.uniform(.15, .85), rng.uniform(.2, .9), rng.uniform(.2, .9), 0.1]
    lab[b, 0, 6:] = d
lab = torch.from_numpy(lab).to(dev)
l0 = tr.train_step(x, lab); torch.cuda.synchronize()
print('first losses', l0.sum(dim=1).tolist())
t0 = time.time()
for _ in range(a.steps):
    l = tr.train_step(x, lab)
torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
print(a.dtype, 'batch %d size %d: %.1f ms/step = %.1f img/s; losses %s; mem %.1f GB' % (a.batch, a.size, dt * 1e3, a.batch / dt, l.sum(dim=1).tolist(), torch.cuda.max_memory_allocated() / 2**30))
