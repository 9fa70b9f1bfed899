"""Random (3x3 conv, following 1x1 conv) pairs through the fused tail (yolo_conv_desc.tail_*) against the two separate launches
on the same buffers: main output and tail output bit-identical, every 256-cout tile variant that takes the shape.
    python tools/fuzz_tail.py <seed> <seconds>      (YOLO_PIPE_PERSIST=8 forces many tiles per persistent block)"""
import sys, os, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from yolo_amd import lib as L
lib = L.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
ncase = nrun = 0
bad = []
t0 = time.time()


def packed(co, ci, k):
    w = torch.from_numpy((rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32)).to(dev)
    wp = torch.empty(lib.yolo_packed_weight_bytes(co, ci, k, L.BF16), dtype=torch.uint8, device=dev)
    L.check(lib.yolo_pack_conv_weights(w.data_ptr(), wp.data_ptr(), co, ci, k, L.BF16, st), 'pack')
    return wp


def sb(co):
    cp = lib.yolo_padded_channels(co)
    s_, b_ = torch.zeros(cp, device=dev), torch.zeros(cp, device=dev)
    s_[:co] = torch.from_numpy(rng.uniform(.5, 1.5, co).astype(np.float32)).to(dev)
    b_[:co] = torch.from_numpy((0.1 * rng.standard_normal(co)).astype(np.float32)).to(dev)
    return s_, b_


while time.time() - t0 < budget:
    cin = int(rng.choice([32, 64, 96, 128, 256])); cout = int(rng.choice([32, 64, 96, 128, 160, 224, 256]))
    stride = int(rng.choice([1, 1, 2])); N = int(rng.choice([1, 2, 3, 7]))
    H = int(rng.integers(2, 60)); W = int(rng.integers(2, 80))
    if N * H * W * max(cin, cout) > 8e6: continue
    res = bool(rng.random() < 0.4)
    tf32 = bool(rng.random() < 0.25)
    tcout = int(rng.choice([90, 18, 30]) if tf32 else rng.choice([16, 32, 64, 96, 128]))
    strided = bool(rng.random() < 0.3)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = torch.from_numpy(rng.standard_normal((N, H, W, cin)).astype(np.float32)).to(dev).bfloat16()
    r = torch.from_numpy(rng.standard_normal((N, Ho, Wo, cout)).astype(np.float32)).to(dev).bfloat16() if res else None
    wp, wp1 = packed(cout, cin, 3), packed(tcout, cout, 1)
    (sc, bi), (sc1, bi1) = sb(cout), sb(tcout)
    ych = cout + 32 if strided else cout
    tpitch = tcout + 6 if tf32 else tcout
    ncase += 1
    for algo in ((10, 16, 18, 9, 17) if stride == 2 else (2, 6, 7)):
        outs = []
        ok = True
        for fused in (False, True):
            ybuf = torch.full((N, Ho, Wo, ych), float('nan'), dtype=torch.bfloat16, device=dev)
            y = ybuf[..., ych - cout:]
            z = torch.full((N, Ho, Wo, tpitch), float('nan'), dtype=torch.float32 if tf32 else torch.bfloat16, device=dev)
            d = L.ConvDesc()
            d.x, d.w_packed, d.scale, d.bias, d.y = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr()
            d.residual = r.data_ptr() if res else None
            d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = N, H, W, cin, cout, 3, stride, L.BF16, 0.1, algo
            d.y_pixel_stride, d.y_batch_stride = ych, Ho * Wo * ych
            if fused:
                d.tail_w_packed, d.tail_scale, d.tail_bias, d.tail_y = wp1.data_ptr(), sc1.data_ptr(), bi1.data_ptr(), z.data_ptr()
                d.tail_cout, d.tail_out_f32, d.tail_slope = tcout, int(tf32), (1.0 if tf32 else 0.1)
                d.tail_y_pixel_stride, d.tail_y_batch_stride = tpitch, Ho * Wo * tpitch
                rc = lib.yolo_conv_fwd(C.byref(d), st)
            else:
                rc = lib.yolo_conv_fwd(C.byref(d), st)
                if rc == 0:
                    d1 = L.ConvDesc()
                    d1.x, d1.w_packed, d1.scale, d1.bias, d1.y = y.data_ptr(), wp1.data_ptr(), sc1.data_ptr(), bi1.data_ptr(), z.data_ptr()
                    d1.N, d1.H, d1.W, d1.Cin, d1.Cout, d1.ksize, d1.stride, d1.dtype = N, Ho, Wo, cout, tcout, 1, 1, L.BF16
                    d1.out_f32, d1.slope, d1.x_pixel_stride = int(tf32), (1.0 if tf32 else 0.1), ych
                    d1.y_pixel_stride, d1.y_batch_stride = tpitch, Ho * Wo * tpitch
                    L.check(lib.yolo_conv_fwd(C.byref(d1), st), '1x1')
            if rc != 0:
                ok = False
                break
            torch.cuda.synchronize()
            outs.append((y.contiguous().clone(), z[..., :tcout].contiguous().clone()))
        if not ok:
            continue
        nrun += 1
        (y0, z0), (y1, z1) = outs
        it = torch.int32 if tf32 else torch.int16
        if not (torch.equal(y0.view(torch.int16), y1.view(torch.int16)) and torch.equal(z0.view(it), z1.view(it))) or torch.isnan(z1.float()).any():
            bad.append((N, cin, H, W, cout, stride, res, tcout, tf32, strided, algo))
            print('MISMATCH', bad[-1], flush=True)
print('cases %d, fused launches checked %d, mismatches %d' % (ncase, nrun, len(bad)))
