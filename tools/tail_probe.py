#!/usr/bin/env python
"""Fused tail 1x1 (yolo_conv_desc.tail_*) against the two separate launches, per shape and 256-cout tile variant.
    python tools/tail_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L
lib = L.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def probe(N, hw, cin, cout, stride, res, tcout, tf32):
    ho = (hw - 1) // stride + 1
    x = torch.randn((N, hw, hw, cin), device=dev).bfloat16()
    def pk(co, ci, k):
        w = torch.randn((co, ci, k, k), device=dev) * 0.05
        wp = torch.empty(lib.yolo_packed_weight_bytes(co, ci, k, 1), dtype=torch.uint8, device=dev)
        lib.yolo_pack_conv_weights(w.data_ptr(), wp.data_ptr(), co, ci, k, 1, st)
        return wp
    wp, wp1 = pk(cout, cin, 3), pk(tcout, cout, 1)
    cp, cp1 = lib.yolo_padded_channels(cout), lib.yolo_padded_channels(tcout)
    sc, bi, sc1, bi1 = torch.ones(cp, device=dev), torch.zeros(cp, device=dev), torch.ones(cp1, device=dev), torch.zeros(cp1, device=dev)
    y = torch.empty((N, ho, ho, cout), device=dev, dtype=torch.bfloat16)
    z = torch.empty((N, ho, ho, tcout), device=dev, dtype=torch.float32 if tf32 else torch.bfloat16)
    r = torch.randn_like(y) if res else None
    d = L.ConvDesc()
    d.x, d.w_packed, d.scale, d.bias, d.y = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr()
    d.residual = r.data_ptr() if res else None
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope = N, hw, hw, cin, cout, 3, stride, 1, 0.1
    d1 = L.ConvDesc()
    d1.x, d1.w_packed, d1.scale, d1.bias, d1.y = y.data_ptr(), wp1.data_ptr(), sc1.data_ptr(), bi1.data_ptr(), z.data_ptr()
    d1.N, d1.H, d1.W, d1.Cin, d1.Cout, d1.ksize, d1.stride, d1.dtype = N, ho, ho, cout, tcout, 1, 1, 1
    d1.out_f32, d1.slope = int(tf32), (1.0 if tf32 else 0.1)
    algos3 = (2, 3, 4, 5, 6, 7, 8, 11, 26) if stride == 1 else (9, 10, 16, 17, 18)
    t3 = {}
    for a in algos3:
        d.algo = a
        if lib.yolo_conv_fwd(C.byref(d), st) == 0:
            t3[a] = timed(lambda: lib.yolo_conv_fwd(C.byref(d), st))
    t1 = {}
    for a in (2, 3, 4, 5, 8, 11, 12, 13, 36, 37, 38, 39, 22, 23, 24):
        d1.algo = a
        if lib.yolo_conv_fwd(C.byref(d1), st) == 0:
            t1[a] = timed(lambda: lib.yolo_conv_fwd(C.byref(d1), st))
    b3, b1 = min(t3, key=t3.get), min(t1, key=t1.get)
    print('   3x3 variants:', ', '.join('%d: %.1f' % (a, t) for a, t in sorted(t3.items())))
    d.algo, d1.algo = b3, b1
    both = timed(lambda: (lib.yolo_conv_fwd(C.byref(d), st), lib.yolo_conv_fwd(C.byref(d1), st)))
    d.tail_w_packed, d.tail_scale, d.tail_bias, d.tail_y = wp1.data_ptr(), sc1.data_ptr(), bi1.data_ptr(), z.data_ptr()
    d.tail_cout, d.tail_out_f32, d.tail_slope = tcout, int(tf32), (1.0 if tf32 else 0.1)
    tf = {}
    for a in ((2, 6, 7) if stride == 1 else (10, 16, 18, 9, 17)):
        d.algo = a
        if lib.yolo_conv_fwd(C.byref(d), st) == 0:
            tf[a] = timed(lambda: lib.yolo_conv_fwd(C.byref(d), st))
    print('N %d %dx%d %d->%d s%d res %d tail %d%s: 3x3 best algo %d %.1f us, 1x1 best algo %d %.1f us, both back to back %.1f us; fused %s'
          % (N, hw, hw, cin, cout, stride, res, tcout, ' f32' if tf32 else '', b3, t3[b3], b1, t1[b1], both,
             ', '.join('algo %d %.1f us' % (a, t) for a, t in sorted(tf.items()))))


for N, s in ((32, 416), (64, 608)):
    probe(N, s // 2, 64, 128, 2, 0, 64, 0)            # stage-1 down-sampling conv + the first block's 1x1
    probe(N, s // 4, 64, 128, 1, 1, 64, 0)            # stage-1 residual 3x3 + the next block's 1x1
    probe(N, s // 8, 128, 256, 1, 1, 128, 0)          # stage-2 residual 3x3 + the next block's 1x1
    probe(N, s // 4, 128, 256, 2, 0, 128, 0)          # stage-2 down-sampling conv + the first block's 1x1
    probe(N, s // 8, 128, 256, 1, 0, 128, 0)          # heads.2 body 3x3 + 1x1
    probe(N, s // 8, 128, 256, 1, 0, 90, 1)           # heads.2 tip + YOLOOutput
