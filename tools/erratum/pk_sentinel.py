#!/usr/bin/env python
"""Is the packed-fp32 corruption a STRAY WRITE?    The victim here has no packed operation: tools/erratum/pk_spin.hip
k_sentinel fills 224 VGPRs of a wave with known values, idles beside the co-runner and checks them.  Co-runners: none; the
synthetic MFMA loop in the real kernel's shape and footprint WITHOUT and WITH `v_mov_b64 v[n:n+1], 0` (spin_dense 9 / 10 --
10 corrupts the packed BatchNorm backward, 9 does not); this library's generic convolution.

    python tools/erratum/pk_sentinel.py [rounds]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from yolo_amd import lib as L


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda:0')
    ship = L.load()
    spin = C.CDLL(os.path.join(ROOT, 'tools', '_build', 'libpk_spin.so'))
    vp = C.c_void_p
    side = torch.cuda.Stream(device=dev)
    spin_out = torch.zeros(1 << 16, device=dev)
    wx = torch.randn((8, 104, 104, 128), device=dev).to(torch.bfloat16)
    cw = torch.randn((128, 128, 3, 3), device=dev)
    cwp = torch.empty(ship.yolo_packed_weight_bytes(128, 128, 3, L.BF16), dtype=torch.uint8, device=dev)
    L.check(ship.yolo_pack_conv_weights(cw.data_ptr(), cwp.data_ptr(), 128, 128, 3, L.BF16, torch.cuda.current_stream().cuda_stream), 'pack')
    cy = torch.empty_like(wx)
    d = L.ConvDesc()
    d.x, d.w_packed, d.y = wx.data_ptr(), cwp.data_ptr(), cy.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = 8, 104, 104, 128, 128, 3, 1, L.BF16, 1.0, 1
    torch.cuda.synchronize()

    def co_conv():
        for _ in range(6):
            L.check(ship.yolo_conv_fwd(C.byref(d), side.cuda_stream), 'conv')

    cos = (('alone', lambda: None),
           ('beside the synthetic MFMA loop, real shape + footprint, NO v_mov_b64 (spin_dense 9)', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 9, 676 * 6, 12, vp(side.cuda_stream))),
           ('beside the synthetic MFMA loop, real shape + footprint, WITH v_mov_b64 zeros (spin_dense 10)', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 10, 676 * 6, 12, vp(side.cuda_stream))),
           ('beside yolo_conv_fwd 3x3, generic kernel', co_conv))
    nblocks, nsent = 2048, 224
    want = (0x3f800000 + torch.arange(nsent, dtype=torch.int64).view(1, nsent, 1) * 64 + torch.arange(64, dtype=torch.int64).view(1, 1, 64)).to(torch.int32).to(dev)
    for name, co in cos:
        total, regs, lanes, vals = 0, {}, {}, {}
        for r in range(rounds):
            out = torch.full((nblocks, nsent, 64), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            co()
            spin.victim_sentinel(vp(out.data_ptr()), nblocks, 40, vp(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            ne = (out != want).nonzero()
            total += int(ne.shape[0])
            for b, k, l in ne[:4096].cpu().tolist():
                regs[k + 16] = regs.get(k + 16, 0) + 1
                lanes[l // 16] = lanes.get(l // 16, 0) + 1
                v = int(out[b, k, l]) & 0xffffffff
                vals[v] = vals.get(v, 0) + 1
        print('%-100s changed registers: %6d in %d rounds' % (name, total, rounds), flush=True)
        if total:
            print('    by register (v16..v239):', dict(sorted(regs.items())))
            print('    by lane quarter (0: lanes 0-15 ... 3: lanes 48-63):', dict(sorted(lanes.items())))
            print('    values found:', {('0x%08x' % k): v for k, v in sorted(vals.items(), key=lambda kv: -kv[1])[:8]})


if __name__ == '__main__':
    main()
