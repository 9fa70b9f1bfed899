#!/usr/bin/env python
"""Phase timing inside conv_pipe_kernel from the instrumented library (make -C yolo_amd/csrc stamp):
per block shader-clock stamps at start / after the prologue / after the K loop / before and after the epilogue /
after its stores have drained, plus HW_ID so blocks can be grouped per CU.
    YOLO_AMD_LIB=yolo_amd/csrc/_stamp/libyolo_amd_stamp.so python tools/stamp_probe.py --cin 128 --cout 256 --hw 76"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolo_amd import lib as L
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=64); ap.add_argument('--hw', type=int, default=76)
ap.add_argument('--cin', type=int, default=128); ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--k', type=int, default=3); ap.add_argument('--algo', type=int, default=8); ap.add_argument('--res', type=int, default=1)
a = ap.parse_args()
lib = L.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
lib.yolo_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]; lib.yolo_debug_read_stamps.restype = C.c_int
x = torch.randn((a.n, a.hw, a.hw, a.cin), device=dev).bfloat16()
w = torch.randn((a.cout, a.cin, a.k, a.k), device=dev) * 0.05
wp = torch.empty(lib.yolo_packed_weight_bytes(a.cout, a.cin, a.k, 1), dtype=torch.uint8, device=dev)
lib.yolo_pack_conv_weights(w.data_ptr(), wp.data_ptr(), a.cout, a.cin, a.k, 1, st)
cp = lib.yolo_padded_channels(a.cout)
sc, bi = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
y = torch.empty((a.n, a.hw, a.hw, a.cout), device=dev, dtype=torch.bfloat16)
r = torch.randn_like(y) if a.res else None
d = L.ConvDesc()
d.x, d.w_packed, d.scale, d.bias, d.y = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr()
d.residual = r.data_ptr() if a.res else None
d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = a.n, a.hw, a.hw, a.cin, a.cout, a.k, 1
d.dtype, d.out_f32, d.slope, d.algo = 1, 0, 0.1, a.algo
for _ in range(3):
    assert lib.yolo_conv_fwd(C.byref(d), st) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); lib.yolo_conv_fwd(C.byref(d), st); e1.record(); torch.cuda.synchronize()
print('kernel %.1f us' % (e0.elapsed_time(e1) * 1e3))
nb, NS = 8192, 16
buf = np.zeros(nb * NS, np.int64)
assert lib.yolo_debug_read_stamps(buf.ctypes.data, nb * NS) == 0
s = buf.reshape(nb, NS)
s = s[s[:, 0] != 0]
print('blocks stamped', len(s))
ph = np.stack([s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4], s[:, 5] - s[:, 0]], 1)
names = ['prologue', 'K loop', 'drain+barrier', 'epilogue issue', 'store drain', 'TOTAL']
for i, n in enumerate(names):
    v = ph[:, i]
    print('%-15s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f cycles' % (n, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
# the epilogue as wave 0 walks it (slots 8-15, conv_epilogue.h): entry -> offsets + residual requests of slab 0 -> per slab:
# transpose written to LDS | its passes (LDS reads, arithmetic, stores issued)
if s[:, 8].any():
    e = s[:, 8:16]
    names2 = ['enter->prefetch0', 'slab0 LDS write', 'slab0 passes', 'slab1 prefetch+write', 'slab1 passes', 'slab2 prefetch+write', 'slab2 passes']
    for i, n in enumerate(names2):
        v = e[:, i + 1] - e[:, i]
        v = v[(e[:, i + 1] != 0) & (e[:, i] != 0)]
        if len(v):
            print('  epi %-22s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f cycles' % (n, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
    print('  epi setup (stamp 3 -> enter): mean %.0f' % (s[:, 8] - s[:, 3]).mean())
wall = s[:, 6]
print('wall span of block starts: %.1f us (100 MHz ticks); first-round blocks (start within 2 us of the first): %d'
      % ((wall.max() - wall.min()) / 100.0, int((wall - wall.min() < 200).sum())))
# per-CU timelines: CU = (XCC_ID, SE_ID, SH_ID, CU_ID)
hw = s[:, 7]
bidx = np.nonzero(buf.reshape(nb, NS)[:, 0] != 0)[0]
key = ((hw >> 32) & 0xf) * 256 + ((hw >> 8) & 0xff)
print('distinct CUs seen: %d' % len(set(key.tolist())))
for k0 in sorted(set(key.tolist()))[:2]:
    m = key == k0
    sel, bi = s[m], bidx[m]
    o = np.argsort(sel[:, 0])
    sel, bi = sel[o], bi[o]
    t0 = sel[0, 0]
    print('CU key %d: %d blocks' % (k0, len(sel)))
    for row, b in list(zip(sel, bi))[:10]:
        print('  block %5d  start %7d  prolog_end %7d  loop_end %7d  epi_start %7d  epi_end %7d  drained %7d' % tuple([b] + list(row[:6] - t0)))
