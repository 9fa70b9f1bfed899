#!/usr/bin/env python
"""One D53 training step in capture mode; every BatchNorm-backward output dy is re-derived on the GPU from the tensors
the kernel read (dz as captured, the saved raw output, the saved statistics) and compared element by element."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolo_amd.net import CarNet
from yolo_amd.train import Trainer
from yolo_amd.spec import darknet53_spec, LEAKY_SLOPE
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tune = sys.argv[2] if len(sys.argv) > 2 else 'measure'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
net = CarNet(darknet53_spec(), dtype='bf16', device=dev, tune=tune).initialize(seed=1234)
tr = Trainer(net, (416, 416))
x = torch.rand((B, 3, 416, 416), generator=torch.Generator().manual_seed(1)).to(dev)
lab = torch.from_numpy(bench.synthetic_labels(B, 3)).to(dev)
bad_total = 0
for it in range(steps):
    cap = {'_poison': 1} if os.environ.get('DY_POISON') else {}
    tr.train_step(x, lab, update=False, capture=cap)
    torch.cuda.synchronize()
    P = tr._last[0]
    for op in P.fwd:
        if op['kind'] != 'conv_bn' or op['c'].name not in cap:
            continue
        c = op['c']
        nn = int(torch.isnan(cap[c.name]['dy'].float()).sum())
        if nn:
            print('step %d %s: %d NaN (unwritten) elements' % (it, c.name, nn), flush=True)
        dz, dy, y = cap[c.name]['dz'].float(), cap[c.name]['dy'].float(), op['yraw'].val.float()
        g, b = net.params[c.name + '.gamma'].float(), net.params[c.name + '.beta'].float()
        mu, inv = op['mean'], op['invstd']
        xh = (y - mu) * inv
        a = g * xh + b
        da = dz * torch.where(a > 0, 1.0, LEAKY_SLOPE)
        n = y.shape[0] * y.shape[1] * y.shape[2]
        k1 = da.double().sum(dim=(0, 1, 2)) / n
        k2 = (da * xh).double().sum(dim=(0, 1, 2)) / n
        ref = g * inv * (da - k1.float() - xh * k2.float())
        err = (dy - ref).abs() / (ref.abs().max() + 1e-30)
        off = (err > 0.02) & (a.abs() > 1e-3)
        if bool(off.any()):
            idx = off.nonzero()
            bad_total += int(off.sum())
            print('step %d %-22s %s: %d elements off; first %s got %s want %s; exact zeros among them: %d' % (
                it, c.name, tuple(y.shape), int(off.sum()), idx[0].tolist(), float(dy[tuple(idx[0])]), float(ref[tuple(idx[0])]),
                int((dy[off] == 0).sum())), flush=True)
            z0 = (dy == 0) & (ref != 0)
            zi = z0.nonzero()
            print('   exact zeros where the reference is not: %d; channels %s; pixels (flat) %s' % (
                int(z0.sum()), sorted({int(i[3]) for i in zi})[:40],
                sorted({int((i[0] * y.shape[1] + i[1]) * y.shape[2] + i[2]) for i in zi})[:40]))
            bypx = {}
            for i in zi.tolist():
                bypx.setdefault((i[0], i[1], i[2]), []).append(i[3])
            for k in sorted(bypx)[:8]:
                print('      pixel', k, 'flat', (k[0] * y.shape[1] + k[1]) * y.shape[2] + k[2], 'channels', bypx[k][:40])
            cs = sorted({int(i[3]) for i in zi})[:4]
            print('   gamma / invstd / mean there:', [(float(g[k]), float(inv[k]), float(mu[k])) for k in cs])
            px = {tuple(i[:3].tolist()) for i in idx[:200]}
            print('   pixels (n, y, x):', sorted(px)[:12], ' channels of the first pixel:',
                  sorted(int(i[3]) for i in idx if tuple(i[:3].tolist()) == tuple(idx[0][:3].tolist()))[:40])
print('dy_check: %d elements off' % bad_total)
