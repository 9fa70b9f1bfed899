#!/usr/bin/env python
"""Which packed instruction form is the victim?    tools/erratum/pk_bisect.py + victim variants narrowed the BatchNorm
corruption to the kernel's prologue; this probe runs ONE packed instruction form at a time (tools/erratum/pk_spin.hip victim_pkform:
v_pk_mov_b32 op_sel:[1,0], v_pk_mul_f32, v_pk_add_f32 with neg modifiers) over two arrays beside this library's kernels (the
co-runners that corrupt the real kernel) and checks every output pair against the instruction's definition.

    python tools/erratum/pk_forms.py [rounds]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from yolo_amd import lib as L


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device('cuda:0')
    ship = L.load()
    spin = C.CDLL(os.path.join(ROOT, 'tools', '_build', 'libpk_spin.so'))
    vp = C.c_void_p
    n = 8 << 20
    g = torch.Generator(device='cpu').manual_seed(5)
    a = torch.randn((n, 2), generator=g).to(dev)
    b = torch.randn((n, 2), generator=g).to(dev)
    want = {0: torch.stack([a[:, 1], b[:, 0]], dim=1), 1: a * b, 2: a - b, 3: a * b, 4: a * b, 5: torch.stack([a[:, 1], a[:, 0]], dim=1), 6: torch.stack([(a * b)[:, 1], (a * b)[:, 0]], dim=1), 7: a * b,
            8: torch.stack([(a * b)[:, 1], (a * b)[:, 0]], dim=1), 9: torch.stack([a[:, 1], a[:, 0]], dim=1)}
    side = torch.cuda.Stream(device=dev)
    wx = torch.randn((8, 104, 104, 128), device=dev).to(torch.bfloat16)
    wdy = torch.randn((8, 104, 104, 128), device=dev).to(torch.bfloat16)
    wdw = torch.zeros((128, 128, 3, 3), device=dev)
    wws = torch.zeros(max(ship.yolo_conv_wgrad_workspace_bytes(128, 128, 3, L.BF16), 16), dtype=torch.uint8, device=dev)
    cw = torch.randn((128, 128, 3, 3), device=dev)
    cwp = torch.empty(ship.yolo_packed_weight_bytes(128, 128, 3, L.BF16), dtype=torch.uint8, device=dev)
    L.check(ship.yolo_pack_conv_weights(cw.data_ptr(), cwp.data_ptr(), 128, 128, 3, L.BF16, torch.cuda.current_stream().cuda_stream), 'pack')
    cy = torch.empty_like(wx)
    d = L.ConvDesc()
    d.x, d.w_packed, d.y = wx.data_ptr(), cwp.data_ptr(), cy.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = 8, 104, 104, 128, 128, 3, 1, L.BF16, 1.0, 1
    ma = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
    torch.cuda.synchronize()

    def co_conv(algo):
        d.algo = algo
        for _ in range(6):
            L.check(ship.yolo_conv_fwd(C.byref(d), side.cuda_stream), 'conv')

    def co_wgrad():
        for _ in range(6):
            ship.yolo_conv_wgrad(wdy.data_ptr(), wx.data_ptr(), wdw.data_ptr(), 8, 104, 104, 128, 128, 3, 1, 0, L.BF16, wws.data_ptr(), side.cuda_stream)

    def co_matmul():
        with torch.cuda.stream(side):
            torch.matmul(ma, ma)

    spin_out = torch.zeros(1 << 16, device=dev)
    only = os.environ.get('PK_FORMS')          # e.g. PK_FORMS=8,9,6,5
    cos = (('alone', lambda: None), ('beside torch.matmul', co_matmul), ('beside yolo_conv_fwd generic', lambda: co_conv(1)),
           ('beside spin_dense 10 (MFMA + v_mov_b64 0)', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 10, 676 * 6, 12, vp(side.cuda_stream))),
           ('beside spin_dense 9 (MFMA only)', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 9, 676 * 6, 12, vp(side.cuda_stream))),
           ('beside yolo_conv_fwd pipelined', lambda: co_conv(4)), ('beside yolo_conv_wgrad row walk', co_wgrad))
    names = {0: 'v_pk_mov_b32 op_sel:[1,0]', 1: 'v_pk_mul_f32', 2: 'v_pk_add_f32 neg_lo/neg_hi', 3: 'v_pk_mul_f32 ; v_mov_b64 src', 4: 'v_pk_mul_f32 ; s_nop 7 ; v_mov_b64', 5: 'v_pk_mov_b32 d, s, s op_sel:[1,0]',
             6: 'v_pk_mul_f32 t ; v_pk_mov_b32 d, t, t', 7: '8 x {load ; pk_mul ; pk_mov swap}',
             8: 'form 6 in a 254-register wave', 9: 'form 5 in a 254-register wave'}
    for form in ([int(x) for x in only.split(',')] if only else (8, 9, 7, 5, 6, 3, 4, 0, 1, 2)):
        for cname, co in cos:
            bad = events = lo = lane48 = zeros = 0
            for r in range(rounds):
                out = torch.full_like(a, float('nan'))
                torch.cuda.synchronize()
                co()
                for _ in range(4):
                    spin.victim_pkform(vp(a.data_ptr()), vp(b.data_ptr()), vp(out.data_ptr()), C.c_longlong(n), form, vp(torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                ne = out.view(torch.int32) != want[form].view(torch.int32)
                k = int(ne.sum())
                if k:
                    events += 1; bad += k
                    idx = ne.nonzero()
                    lo += int((idx[:, 1] == 0).sum())
                    lane48 += int(((idx[:, 0] % 64) >= 48).sum())
                    zeros += int((out[ne] == 0).sum())
            print('%-34s %-34s rounds with a mismatch %3d / %d, elements %7d (low element %d, lanes 48-63 %d, exact zeros %d)' % (
                names[form], cname, events, rounds, bad, lo, lane48, zeros), flush=True)


if __name__ == '__main__':
    main()
