#!/usr/bin/env python
"""Where does the packed-fp32 corruption of the BatchNorm backward come from?  (csrc/Makefile NOPK, DESIGN 4.2, round-2
verdict item 7.)  tools/erratum/pk_repro.hip -- a synthetic packed-VALU kernel beside a synthetic MFMA spinner -- does NOT
reproduce it, so this script isolates the REAL kernel outside the Trainer: yolo_bn_train_bwd_pp from a library whose
train.hip was built WITH the packed operations (`make -C yolo_amd/csrc pk` -> yolo_amd/csrc/_ab/libyolo_pk.so) runs
on stream 1 on one fixed input while stream 2 runs, in turn: nothing; a bf16 torch.matmul (hipBLASLt's MFMA kernel: not
this repository's code, its own buffers only); synthetic one-feature spinners (tools/erratum/pk_spin.hip: MFMA, plain and
transposing LDS reads, fp32 atomics, LDS-DMA); this library's forward convolution (generic register-staged kernel and
pipelined LDS-DMA kernel) and its weight gradients (GEMM, strip, row walk) -- all on buffers of their own.  Every dy is
compared bit for bit with the same call of the shipped (no packed operations) library run alone.

    python tools/erratum/pk_bisect.py [rounds]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from yolo_amd import lib as L

PK = os.environ.get('PK_LIB') or os.path.join(L.CSRC, '_ab', 'libyolo_pk.so')        # (PK_LIB: a variant build of the victim)
TRIG = os.environ.get('TRIG_LIB')        # (a variant build of the co-running convolution, tools/erratum/pk_trigger.sh)
ONLY = os.environ.get('PK_ONLY')         # (only the co-runners whose name contains this)


def load(path):
    lib = C.CDLL(path)
    for name, (res, args) in L.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device('cuda:0')
    ship, pk = L.load(), load(PK)
    trig = load(TRIG) if TRIG else ship
    g = torch.Generator(device='cpu').manual_seed(3)
    results = {}
    for (N, H, W, Cc) in ((4, 208, 208, 64), (64, 52, 52, 256)):
        y = torch.randn((N, H, W, Cc), generator=g).to(dev).to(torch.bfloat16)
        dz = (0.01 * torch.randn((N, H, W, Cc), generator=g)).to(dev).to(torch.bfloat16)
        yf = y.float()
        mean = yf.mean(dim=(0, 1, 2)).contiguous()
        invstd = (1.0 / torch.sqrt(yf.var(dim=(0, 1, 2), unbiased=False) + 1e-5)).contiguous()
        gamma = (0.5 + torch.rand(Cc, generator=g)).to(dev)
        beta = (0.1 * torch.randn(Cc, generator=g)).to(dev)
        npix = N * H * W
        ws = [torch.zeros(2 * 2048, dtype=torch.float64, device=dev) for _ in range(2)]
        dgam, dbet = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)

        def bn(lib, out, stream, k):
            ws[k & 1].zero_()
            rc = lib.yolo_bn_train_bwd_pp(dz.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                          beta.data_ptr(), out.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), ws[k & 1].data_ptr(),
                                          ws[(k & 1) ^ 1].data_ptr(), 4096, npix, Cc, 0.1, L.BF16, stream)
            assert rc == 0, rc

        ref = torch.empty_like(y)
        bn(ship, ref, torch.cuda.current_stream().cuda_stream, 0)
        torch.cuda.synchronize()
        ref2 = torch.empty_like(y)
        bn(pk, ref2, torch.cuda.current_stream().cuda_stream, 1)
        torch.cuda.synchronize()
        same_alone = bool(torch.equal(ref.view(torch.int16), ref2.view(torch.int16)))
        # co-runners (all on buffers of their own)
        a = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
        b = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
        wx = torch.randn((8, 104, 104, 128), device=dev).to(torch.bfloat16)
        wdy = torch.randn((8, 104, 104, 128), device=dev).to(torch.bfloat16)
        wdw = torch.zeros((128, 128, 3, 3), device=dev)
        wdw1 = torch.zeros((128, 128, 1, 1), device=dev)
        wx32 = torch.randn((8, 104, 104, 32), device=dev).to(torch.bfloat16)
        wdw32 = torch.zeros((128, 32, 3, 3), device=dev)
        wws = torch.zeros(max(ship.yolo_conv_wgrad_workspace_bytes(128, 128, 3, L.BF16), 16), dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream(device=dev)
        spin = C.CDLL(os.path.join(ROOT, 'tools', '_build', 'libpk_spin.so'))
        spin_src = torch.full((4 << 20,), 0.0078125, device=dev, dtype=torch.bfloat16)
        spin_out = torch.zeros(1 << 16, device=dev)
        vp = C.c_void_p
        # this library's forward convolution (LDS-DMA + MFMA, no transposing reads, no atomics)
        cw = torch.randn((128, 128, 3, 3), device=dev)
        cwp = torch.empty(ship.yolo_packed_weight_bytes(128, 128, 3, L.BF16), dtype=torch.uint8, device=dev)
        L.check(ship.yolo_pack_conv_weights(cw.data_ptr(), cwp.data_ptr(), 128, 128, 3, L.BF16, torch.cuda.current_stream().cuda_stream), 'pack')
        cy = torch.empty_like(wx)
        d = L.ConvDesc()
        d.x, d.w_packed, d.y = wx.data_ptr(), cwp.data_ptr(), cy.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = 8, 104, 104, 128, 128, 3, 1, L.BF16, 1.0, 4
        torch.cuda.synchronize()

        def co_none():
            pass

        def co_matmul():
            with torch.cuda.stream(side):
                for _ in range(2):
                    torch.matmul(a, b)

        def wg(x_, dw_, cin, k):
            for _ in range(6):
                ship.yolo_conv_wgrad(wdy.data_ptr(), x_.data_ptr(), dw_.data_ptr(), 8, 104, 104, cin, 128, k, 1, 0, L.BF16,
                                     wws.data_ptr(), side.cuda_stream)

        def co_conv(algo):
            d.algo = algo
            for _ in range(6):
                L.check(trig.yolo_conv_fwd(C.byref(d), side.cuda_stream), 'conv')

        cos = (('alone', co_none), ('beside torch.matmul (hipBLASLt)', co_matmul),
               ('beside a synthetic MFMA spinner', lambda: spin.spin_mfma(vp(spin_src.data_ptr()), vp(spin_out.data_ptr()), 3000, C.c_longlong(spin_src.numel() // 8), vp(side.cuda_stream))),
               ('beside a synthetic ds_read_b64 spinner', lambda: spin.spin_lds(vp(spin_out.data_ptr()), 4000, 0, vp(side.cuda_stream))),
               ('beside a synthetic ds_read_b64_tr_b16 spinner', lambda: spin.spin_lds(vp(spin_out.data_ptr()), 4000, 1, vp(side.cuda_stream))),
               ('beside a synthetic fp32-atomics spinner', lambda: spin.spin_atomic(vp(spin_out.data_ptr()), 300, spin_out.numel(), vp(side.cuda_stream))),
               ('beside a synthetic LDS-DMA spinner (m0 + global_load_lds_dwordx4)', lambda: spin.spin_dma(vp(spin_src.data_ptr()), vp(spin_out.data_ptr()), 1500, C.c_longlong(spin_src.numel() * 2), vp(side.cuda_stream))),
               ('beside a synthetic turnover spinner: short MFMA blocks (1 wave)', lambda: spin.spin_turnover(vp(spin_src.data_ptr()), vp(spin_out.data_ptr()), 0, 400000, 6, C.c_longlong(spin_src.numel() // 8), vp(side.cuda_stream))),
               ('beside a synthetic turnover spinner: short MFMA blocks (4 waves)', lambda: spin.spin_turnover(vp(spin_src.data_ptr()), vp(spin_out.data_ptr()), 3, 100000, 6, C.c_longlong(spin_src.numel() // 8), vp(side.cuda_stream))),
               ('beside a synthetic turnover spinner: short MFMA blocks + 64 KB LDS + barrier', lambda: spin.spin_turnover(vp(spin_src.data_ptr()), vp(spin_out.data_ptr()), 1, 60000, 6, C.c_longlong(spin_src.numel() // 8), vp(side.cuda_stream))),
               ('beside a synthetic turnover spinner: short VALU-only blocks (200 zeroed registers)', lambda: spin.spin_turnover(vp(spin_src.data_ptr()), vp(spin_out.data_ptr()), 2, 800000, 0, C.c_longlong(spin_src.numel() // 8), vp(side.cuda_stream))),
               ('beside a synthetic DENSE MFMA spinner (16 accumulators, operands in registers), long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 0, 512, 4000, vp(side.cuda_stream))),
               ('beside a synthetic DENSE MFMA spinner, operands from LDS, long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 1, 512, 4000, vp(side.cuda_stream))),
               ('beside a synthetic DENSE MFMA spinner, operands from LDS + barrier, long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 2, 512, 4000, vp(side.cuda_stream))),
               ('beside a synthetic DENSE MFMA spinner, operands from LDS + barrier, short blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 2, 40000, 50, vp(side.cuda_stream))),
               ('beside a synthetic DENSE MFMA spinner, LDS reads + writes + two barriers, long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 3, 512, 4000, vp(side.cuda_stream))),
               ('beside a synthetic DENSE MFMA spinner, LDS reads + writes + two barriers, short blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 3, 40000, 50, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner, each MFMA on JUST-READ LDS fragments (read, wait, use), long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 4, 512, 3000, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner, each MFMA on JUST-READ LDS fragments, next reads in flight, long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 5, 512, 3000, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner, each MFMA on JUST-READ LDS fragments, next reads in flight, short blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 5, 40000, 40, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner in the REAL LOOP SHAPE (4 accumulators, swizzled 64-byte-row reads, 2 barriers), long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 6, 512, 600, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner in the REAL LOOP SHAPE, short blocks (676 x 12 steps)', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 6, 676 * 6, 12, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner in the REAL LOOP SHAPE, reads with 4-way BANK CONFLICTS, long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 7, 512, 600, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner in the REAL LOOP SHAPE, reads with 4-way BANK CONFLICTS, short blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 7, 676 * 6, 12, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner in the REAL LOOP SHAPE, rows wrapping after 26 (few BANK CONFLICTS), long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 8, 512, 600, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner in the REAL LOOP SHAPE, rows wrapping after 26 (few BANK CONFLICTS), short blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 8, 676 * 6, 12, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner in the REAL LOOP SHAPE with the real FOOTPRINT (48 KB LDS, 148 + 64 registers), long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 9, 512, 600, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner in the REAL LOOP SHAPE with the real FOOTPRINT (48 KB LDS, 148 + 64 registers), short blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 9, 676 * 6, 12, vp(side.cuda_stream))),
               ('beside a synthetic VALU spinner: v_mov_b64 v[n:n+1], 0 (no MFMA, no LDS)', lambda: spin.spin_mov64(vp(spin_out.data_ptr()), 0, 768, 20000, vp(side.cuda_stream))),
               ('beside a synthetic VALU spinner: v_mov_b64 v[n:n+1], 1.0', lambda: spin.spin_mov64(vp(spin_out.data_ptr()), 1, 768, 20000, vp(side.cuda_stream))),
               ('beside a synthetic VALU spinner: the same zeros by v_mov_b32 pairs (control for v_mov_b64)', lambda: spin.spin_mov64(vp(spin_out.data_ptr()), 2, 768, 20000, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner, REAL LOOP SHAPE + FOOTPRINT + v_mov_b64 zero fills, long blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 10, 512, 600, vp(side.cuda_stream))),
               ('beside a synthetic MFMA spinner, REAL LOOP SHAPE + FOOTPRINT + v_mov_b64 zero fills, short blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 10, 676 * 6, 12, vp(side.cuda_stream))),
               ('beside a synthetic DENSE MFMA spinner (operands in registers), short blocks', lambda: spin.spin_dense(vp(spin_out.data_ptr()), 0, 40000, 50, vp(side.cuda_stream))),
               ('beside yolo_conv_fwd 3x3, generic kernel (register-staged MFMA)', lambda: co_conv(1)),
               ('beside yolo_conv_fwd 3x3, pipelined kernel (LDS-DMA + MFMA)', lambda: co_conv(4)),
               ('beside yolo_conv_wgrad 1x1 (GEMM / per-tap kernel)', lambda: wg(wx, wdw1, 128, 1)),
               ('beside yolo_conv_wgrad 3x3 Cin 32 (strip kernel)', lambda: wg(wx32, wdw32, 32, 3)),
               ('beside yolo_conv_wgrad 3x3 (row walk)', lambda: wg(wx, wdw, 128, 3)))
        if ONLY:
            cos = tuple(c for c in cos if ONLY in c[0])
        for lib_name, lib in (('packed', pk),) if ONLY else (('packed', pk), ('shipped', ship)):
            for co_name, co in cos:
                bad = zeros = events = 0
                lanes = {}
                pix_lo, pix_hi = 1 << 60, -1
                for r in range(rounds):
                    out = torch.full_like(y, float('nan'))
                    torch.cuda.synchronize()
                    co()
                    bn(lib, out, torch.cuda.current_stream().cuda_stream, r)
                    torch.cuda.synchronize()
                    ne = out.view(torch.int16) != (ref2 if lib is pk else ref).view(torch.int16)      # (each library against itself, alone)
                    k = int(ne.sum())
                    if k:
                        events += 1
                        bad += k
                        zeros += int((out[ne] == 0).sum())
                        allidx = ne.reshape(-1).nonzero().flatten()
                        pix_lo, pix_hi = min(pix_lo, int(allidx.min()) // Cc), max(pix_hi, int(allidx.max()) // Cc)
                        idx = allidx[:64].cpu().tolist()
                        for i in idx:
                            lanes[(i // 8) % 64] = lanes.get((i // 8) % 64, 0) + 1
                key = '%dx%dx%dx%d %s BatchNorm backward %s' % (N, H, W, Cc, lib_name, co_name)
                results[key] = (events, bad, zeros)
                print('%-104s rounds with a mismatch %3d / %d, elements %6d (exact zeros %6d)%s' % (
                    key, events, rounds, bad, zeros, ('  pixels %d..%d of %d' % (pix_lo, pix_hi, npix)) if lanes else ''), flush=True)
        print('   (packed library alone == shipped library alone: %s)' % same_alone, flush=True)
    return results


if __name__ == '__main__':
    main()
