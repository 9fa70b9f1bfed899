#!/usr/bin/env python
"""Does an MFMA-bound convolution overlap with an HBM-bound BatchNorm pass when they run on two streams?
Times conv alone, BN (train forward: reduce + apply) alone, both back to back on one stream, and both on two streams."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L

lib = L.load(); dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
p = lambda t: t.data_ptr()
for (hw, cin, cout) in [(52, 128, 256), (26, 256, 512), (104, 64, 128)]:
    x = torch.randn((B, hw, hw, cin), device=dev).bfloat16()
    y = torch.empty((B, hw, hw, cout), device=dev, dtype=torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    wp = torch.empty(lib.yolo_packed_weight_bytes(cout, cin, 3, L.BF16), dtype=torch.uint8, device=dev)
    L.check(lib.yolo_pack_conv_weights(p(w), p(wp), cout, cin, 3, L.BF16, torch.cuda.current_stream().cuda_stream), 'pack')
    d = L.ConvDesc()
    d.x, d.w_packed, d.scale, d.bias, d.y = p(x), p(wp), None, None, p(y)
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = B, hw, hw, cin, cout, 3, 1, L.BF16, 1.0, 0
    # the BN pass works on ANOTHER tensor of the same size (as a pipelined half-batch would)
    y2 = torch.randn((B, hw, hw, cout), device=dev).bfloat16(); z2 = torch.empty_like(y2)
    g, b = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    mean, inv, rm, rv = (torch.zeros(cout, device=dev) for _ in range(4))
    ws = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
    npix = B * hw * hw
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    conv = lambda s: L.check(lib.yolo_conv_fwd(C.byref(d), s.cuda_stream), 'conv')
    bn = lambda s: L.check(lib.yolo_bn_train_fwd(p(y2), p(g), p(b), None, p(z2), p(mean), p(inv), p(rm), p(rv), p(ws), npix, cout,
                                                 1e-5, 0.9, 0.1, L.BF16, s.cuda_stream), 'bn')

    def timed(fn, iters=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(iters):
            fn()
        s1.wait_stream(s2)
        e1.record(s1); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    def both_two():
        s2.wait_stream(s1)          # (no dependency in this probe beyond start-of-iteration ordering)
        conv(s1); bn(s2)
        s1.wait_stream(s2)
    tc = timed(lambda: conv(s1)); tb = timed(lambda: bn(s1))
    tser = timed(lambda: (conv(s1), bn(s1))); tpar = timed(both_two)
    print('%3d^2 %4d->%4d bs %d: conv %6.1f us, bn %6.1f us, serial %6.1f, two streams %6.1f' % (hw, cin, cout, B, tc, tb, tser, tpar), flush=True)
