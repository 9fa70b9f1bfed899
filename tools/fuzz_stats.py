"""Random convolution shapes with the BatchNorm statistics epilogue (yolo_conv_desc.stats, mode 1) on every variant that offers it:
the mean / invstd that yolo_bn_train_fwd_partials derives from the partial rows against the statistics of the stored output
itself (torch, fp64 on the GPU).    python tools/fuzz_stats.py <seed> <seconds>"""
import sys, os, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo_amd import lib as L
from yolo_amd.net import CarNet
lib = L.load(); dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
st = torch.cuda.current_stream().cuda_stream
ncase = nrun = 0
bad = []
t0 = time.time()
while time.time() - t0 < budget:
    k = int(rng.choice([1, 3])); s = int(rng.choice([1, 1, 2])) if k == 3 else 1
    cin = int(rng.choice([8, 16, 32, 64, 128, 256, 512])); cout = int(rng.choice([8, 32, 48, 64, 128, 256, 512]))
    N = int(rng.choice([1, 2, 3, 5])); H = int(rng.integers(1, 40)); W = int(rng.integers(1, 70))
    if rng.random() < 0.2: H = W = int(rng.choice([13, 19, 26, 38, 52]))
    if N * H * W * max(cin, cout) > 6e6: continue
    pad = k // 2; Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    npix = N * Ho * Wo
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((N, H, W, cin), device=dev, generator=g).to(torch.bfloat16)
    w = torch.randn((cout, cin, k, k), device=dev, generator=g) / float(np.sqrt(cin * k * k))
    wp = torch.empty(int(lib.yolo_packed_weight_bytes(cout, cin, k, L.BF16)), dtype=torch.uint8, device=dev)
    if lib.yolo_pack_conv_weights(w.data_ptr(), wp.data_ptr(), cout, cin, k, L.BF16, st) != 0: continue
    cp = lib.yolo_padded_channels(cout)
    gamma = torch.ones(cout, device=dev); beta = torch.zeros(cout, device=dev)
    ctx = (N, cin, H, W, cout, k, s)
    ncase += 1
    for algo in (0,) + tuple(CarNet.ALGOS):
        y = torch.full((N, Ho, Wo, cout), float('nan'), dtype=torch.bfloat16, device=dev)
        d = L.ConvDesc()
        d.x, d.w_packed, d.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = N, H, W, cin, cout, k, s, L.BF16, 1.0, algo
        d.stats, d.stats_mode = 1, 1
        rows = lib.yolo_conv_stats_rows(C.byref(d))
        if rows <= 0: continue
        part = torch.full((rows, 2, cp), float('nan'), device=dev)
        d.stats = part.data_ptr()
        if lib.yolo_conv_fwd(C.byref(d), st) != 0:
            bad.append(('stats rows promised, launch refused', algo, ctx)); continue
        m1, i1 = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        z1 = torch.empty_like(y); rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        wa, wb = torch.zeros(2 * cout, dtype=torch.float64, device=dev), torch.zeros(2 * cout, dtype=torch.float64, device=dev)
        rc = lib.yolo_bn_train_fwd_partials(part.data_ptr(), rows, cp, y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, z1.data_ptr(), m1.data_ptr(),
                                            i1.data_ptr(), rm.data_ptr(), rv.data_ptr(), wa.data_ptr(), wb.data_ptr(), 2 * cout, npix, cout, 1e-5, 0.9, 0.1, L.BF16, st)
        torch.cuda.synchronize()
        nrun += 1
        if rc != 0: bad.append(('partials rc %d' % rc, algo, ctx)); continue
        if torch.isnan(y.float()).any(): bad.append(('NaN (unwritten) output', algo, ctx)); continue
        yd = y.double().reshape(-1, cout)
        mean = yd.mean(dim=0); var = yd.var(dim=0, unbiased=False); inv = 1.0 / torch.sqrt(var + 1e-5)
        em = float((m1.double() - mean).abs().max()); ei = float(((i1.double() - inv).abs() / inv).max())
        if not (em < 1e-4 * (1 + float(mean.abs().max())) and ei < 2e-3):
            bad.append(('mean err %.3g, invstd rel err %.3g' % (em, ei), algo, ctx))
print('cases %d, kernel runs with the epilogue %d, problems %d' % (ncase, nrun, len(bad)))
for b in bad[:20]: print('  ', b)
