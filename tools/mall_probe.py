#!/usr/bin/env python
"""Streaming rates inside and beyond the 256 MB MALL: torch copy / scale kernels and this library's BatchNorm apply pass
(forward, bf16) on tensors of 16 MB ... 1 GB, back to back (so a tensor that fits is read from the cache it was left in).
    python tools/mall_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L
lib = L.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


Cc = 256
for mb in (16, 32, 64, 128, 256, 512, 1024):
    npix = mb * (1 << 20) // (Cc * 2)
    y = torch.randn((npix, Cc), device=dev).bfloat16()
    z = torch.empty_like(y)
    g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    mean, inv = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    ws = [torch.zeros(2 * Cc, dtype=torch.float64, device=dev) for _ in range(2)]
    t_copy = timed(lambda: z.copy_(y))
    t_scale = timed(lambda: torch.mul(y, 2.0, out=z))
    t_read = timed(lambda: y.sum())

    def bn():
        ws[0].zero_()
        L.check(lib.yolo_bn_train_fwd_pp(y.data_ptr(), g.data_ptr(), b.data_ptr(), None, z.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                         rm.data_ptr(), rv.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), 2 * Cc, npix, Cc, 1e-5, 0.9, 0.1, L.BF16, st), 'bn')
    t_bn = timed(bn)
    by = y.numel() * 2
    print('%5d MB tensor: copy %.2f TB/s, scale %.2f TB/s, reduction (read only) %.2f TB/s, BatchNorm fwd (reduce + apply: 3 x) %.2f TB/s  (%.1f us)'
          % (mb, 2 * by / t_copy / 1e12, 2 * by / t_scale / 1e12, by / t_read / 1e12, 3 * by / t_bn / 1e12, t_bn * 1e6))
