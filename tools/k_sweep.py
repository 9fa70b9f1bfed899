#!/usr/bin/env python
"""Time one conv variant over a sweep of Cin (K = 9*Cin or Cin) at a fixed output tensor: the intercept of the
time-vs-K line is the per-tile fixed cost (prologue + epilogue), the slope the main-loop rate."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=64); ap.add_argument('--hw', type=int, default=76)
ap.add_argument('--cout', type=int, default=256); ap.add_argument('--k', type=int, default=3)
ap.add_argument('--algos', default='8,6,4,2'); ap.add_argument('--cins', default='64,128,256,512,1024')
a = ap.parse_args()
lib = L.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
for algo in [int(v) for v in a.algos.split(',')]:
    for res in (0, 1):
        row = []
        for cin in [int(v) for v in a.cins.split(',')]:
            x = torch.randn((a.n, a.hw, a.hw, cin), device=dev).bfloat16()
            w = torch.randn((a.cout, cin, a.k, a.k), device=dev) * 0.05
            wp = torch.empty(lib.yolo_packed_weight_bytes(a.cout, cin, a.k, 1), dtype=torch.uint8, device=dev)
            lib.yolo_pack_conv_weights(w.data_ptr(), wp.data_ptr(), a.cout, cin, a.k, 1, st)
            cp = lib.yolo_padded_channels(a.cout)
            sc, bi = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
            y = torch.empty((a.n, a.hw, a.hw, a.cout), device=dev, dtype=torch.bfloat16)
            r = torch.randn_like(y) if res else None
            d = L.ConvDesc()
            d.x, d.w_packed, d.scale, d.bias, d.y = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr()
            d.residual = r.data_ptr() if res else None
            d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = a.n, a.hw, a.hw, cin, a.cout, a.k, 1
            d.dtype, d.out_f32, d.slope, d.algo = 1, 0, 0.1, algo
            if lib.yolo_conv_fwd(C.byref(d), st) != 0:
                row.append('  --  '); continue
            for _ in range(3): lib.yolo_conv_fwd(C.byref(d), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): lib.yolo_conv_fwd(C.byref(d), st)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            fl = 2.0 * a.n * a.hw * a.hw * cin * a.cout * a.k * a.k
            row.append('%6.1f us %5.0f TF' % (us, fl / us / 1e6))
        print('algo %2d res %d | %s' % (algo, res, ' | '.join(row)), flush=True)
