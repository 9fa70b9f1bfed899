#!/usr/bin/env python
"""Run ONE conv variant on one layer shape a few times (for rocprofv3 --pmc passes on a single kernel).
    python tools/one_layer.py --n 32 --hw 13 --cin 1024 --cout 512 --k 1 --algo 11"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=32); ap.add_argument('--hw', type=int, default=13)
ap.add_argument('--cin', type=int, default=1024); ap.add_argument('--cout', type=int, default=512)
ap.add_argument('--k', type=int, default=1); ap.add_argument('--s', type=int, default=1); ap.add_argument('--algo', type=int, default=11); ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
x = torch.randn((a.n, a.hw, a.hw, a.cin), device=dev).bfloat16()
w = torch.randn((a.cout, a.cin, a.k, a.k), device=dev) * 0.05
wp = torch.empty(lib.yolo_packed_weight_bytes(a.cout, a.cin, a.k, 1), dtype=torch.uint8, device=dev)
lib.yolo_pack_conv_weights(w.data_ptr(), wp.data_ptr(), a.cout, a.cin, a.k, 1, st)
cp = lib.yolo_padded_channels(a.cout)
sc, bi = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
ho = (a.hw + 2 * (a.k // 2) - a.k) // a.s + 1
y = torch.empty((a.n, ho, ho, a.cout), device=dev, dtype=torch.bfloat16)
d = L.ConvDesc()
d.x, d.w_packed, d.scale, d.bias, d.y = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr()
d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = a.n, a.hw, a.hw, a.cin, a.cout, a.k, a.s
d.dtype, d.out_f32, d.slope, d.algo = 1, 0, 0.1, a.algo
for _ in range(a.iters):
    assert lib.yolo_conv_fwd(C.byref(d), st) == 0
torch.cuda.synchronize()
