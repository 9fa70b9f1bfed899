#!/usr/bin/env python
"""Time every conv variant on one layer shape (back-to-back launches): which tile / pipeline wins and by how much.
    python tools/algo_times.py --n 32 --hw 13 --cin 1024 --cout 512 --k 1"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=32); ap.add_argument('--hw', type=int, default=13)
ap.add_argument('--cin', type=int, default=1024); ap.add_argument('--cout', type=int, default=512)
ap.add_argument('--k', type=int, default=1); ap.add_argument('--s', type=int, default=1); ap.add_argument('--f32out', type=int, default=0); ap.add_argument('--res', type=int, default=0); ap.add_argument('--iters', type=int, default=50)
a = ap.parse_args()
dev = torch.device('cuda:0')
net = CarNet(darknet53_spec(), device=dev)
lib, st = net._lib, L.stream_ptr()
x = torch.randn((a.n, a.hw, a.hw, a.cin), device=dev).bfloat16()
w = torch.randn((a.cout, a.cin, a.k, a.k), device=dev) * 0.05
wp = torch.empty(lib.yolo_packed_weight_bytes(a.cout, a.cin, a.k, 1), dtype=torch.uint8, device=dev)
lib.yolo_pack_conv_weights(w.data_ptr(), wp.data_ptr(), a.cout, a.cin, a.k, 1, st)
cp = lib.yolo_padded_channels(a.cout)
sc, bi = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
ho = (a.hw + 2 * (a.k // 2) - a.k) // a.s + 1
y = torch.empty((a.n, ho, ho, a.cout), device=dev, dtype=torch.float32 if a.f32out else torch.bfloat16)
r = torch.randn_like(y) if a.res else None
d = L.ConvDesc()
d.x, d.w_packed, d.scale, d.bias, d.y = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr()
d.residual = r.data_ptr() if a.res else None
d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = a.n, a.hw, a.hw, a.cin, a.cout, a.k, a.s
d.dtype, d.out_f32, d.slope = 1, a.f32out, (1.0 if a.f32out else 0.1)
out = []
for algo in net.ALGOS:
    d.algo = algo
    if lib.yolo_conv_fwd(C.byref(d), st) != 0:
        continue
    for _ in range(5): lib.yolo_conv_fwd(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): lib.yolo_conv_fwd(C.byref(d), st)
    e1.record(); e1.synchronize()
    out.append((e0.elapsed_time(e1) / a.iters * 1e3, algo))
fl = 2.0 * a.n * ho * ho * a.cin * a.cout * a.k * a.k
for t, algo in sorted(out):
    print('algo %4d  %7.1f us  %6.0f TF' % (algo, t, fl / t / 1e6))
