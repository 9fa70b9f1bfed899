"""Diagnostic: gradients of a batch made of R copies of a 2-image batch vs R x the 2-image gradients, per layer.
    python tools/rep_batch_diag.py [f32|bf16] [R] [auto|measure]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import graph as og, train as ot
from yolo_amd.net import CarNet
from yolo_amd.train import Trainer

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
R = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tune = sys.argv[3] if len(sys.argv) > 3 else 'auto'
dev = torch.device('cuda:0')
spec, size = og.spec_d53(), (416, 416)
g = og.build_graph(spec)
P = og.init_params(g, seed=0, bn='random')
x2 = np.random.default_rng(2).random((2, 3) + size, dtype=np.float32)
lab2 = ot.synthetic_labels(2, seed=3, render_rate=0.0, num_class=24)
net = CarNet(spec, dtype=dtype, device=dev, tune=tune).load_params(P)
tr = Trainer(net, size)
x = torch.from_numpy(np.tile(x2, (R, 1, 1, 1))).to(dev)
lab = torch.from_numpy(np.tile(lab2, (R, 1, 1))).to(dev)
l64 = tr.train_step(x, lab, update=False).cpu().numpy()
m64 = tr.merged_logits()[:2].clone()
g64 = {n: v.clone() for n, v in tr.grads().items()}
l2 = tr.train_step(x[:2], lab[:2], update=False).cpu().numpy()
m2 = tr.merged_logits()[:2].clone()
print('losses rep[:2]', l64[:, :2].sum(1), 'B=2', l2.sum(1))
print('logits max diff', float((m64 - m2).abs().max()), 'scale', float(m2.abs().max()))
for n in tr.names:
    a = g64[n].double().flatten()
    b = tr.grads()[n].double().flatten() * R
    print('%-28s cos %.4f ratio %.4f relL2 %.3g' % (n, float(a @ b / (a.norm() * b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30)),
                                                   float((a - b).norm() / (b.norm() + 1e-30))))
