"""Random convolution shapes through yolo_conv_fwd, every variant that accepts the shape, against the fp32 torch reference on
bf16-rounded operands (tests/util.py ref_conv)."""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from yolo_amd import lib as L
from yolo_amd.net import CarNet
from util import run_conv, ref_conv, ref_conv_split
lib = L.load(); dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ALGOS = (0,) + tuple(CarNet.ALGOS)
ncase = nrun = 0
bad = []
t0 = time.time()
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 120:
    k = int(rng.choice([1, 3])); s = int(rng.choice([1, 1, 2])) if k == 3 else 1
    cin = int(rng.choice([8, 16, 24, 32, 40, 64, 72, 96, 128, 192, 256, 320, 512]))
    cout = int(rng.choice([8, 18, 30, 32, 48, 64, 90, 96, 128, 160, 256, 384, 512]))
    N = int(rng.choice([1, 2, 3, 5]))
    H = int(rng.integers(1, 40)); W = int(rng.integers(1, 70))
    if rng.random() < 0.2: H, W = int(rng.choice([13, 19, 26, 38, 52])), int(rng.choice([13, 19, 26, 38, 52]))
    if N * H * W * max(cin, cout) > 6e6: continue
    res = bool(rng.random() < 0.3) and s == 1
    slope = float(rng.choice([0.1, 0.0, 1.0]))
    x = rng.standard_normal((N, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    scale = (0.5 + rng.random(cout)).astype(np.float32); bias = (0.2 * rng.standard_normal(cout)).astype(np.float32)
    pad = k // 2; Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    r = rng.standard_normal((N, cout, Ho, Wo)).astype(np.float32) if res else None
    u = rng.random()
    dt = 'f32' if u < 0.15 else ('bf16x3' if u < 0.4 else 'f16x3' if u < 0.65 else 'bf16')      # (split types, round 6: against the split arithmetic restated on the CPU)
    if dt in ('bf16x3', 'f16x3'):
        if cin % 8 or cout % 8: continue
        want = ref_conv_split(x, w, scale, bias, s, slope, residual=r, rdt=torch.bfloat16 if dt == 'bf16x3' else torch.float16)
    else:
        want = ref_conv(x, w, scale, bias, s, slope, residual=r, bf16=(dt == 'bf16'))
    ncase += 1
    for algo in ALGOS:
        try:
            got = run_conv(lib, dev, x, w, scale, bias, s, slope, dt, residual=r, algo=algo, expect_rc=None)
        except Exception as e:
            bad.append(('EXC', algo, (N, cin, H, W, cout, k, s, res, slope), repr(e)[:120])); continue
        if got is None:
            if algo == 0 and (cout % 8 == 0 or dt == 'f32') and not (dt == 'f32' and cout % 4): bad.append(('algo 0 refused', dt, (N, cin, H, W, cout, k, s, res, slope)))
            continue
        nrun += 1
        if np.isnan(got).any():
            bad.append(('NaN (unwritten output)', algo, (N, cin, H, W, cout, k, s, res, slope))); continue
        err = np.abs(got - want); tol = {'bf16': 0.02, 'f32': 2e-4, 'bf16x3': 4e-5, 'f16x3': 4e-5}[dt] * np.maximum(np.abs(want), 1.0)       # ~2 bf16 ulps of the result + slack for order
        if (err > tol).any():
            bad.append(('mismatch %.3g' % float(err.max()), dt, algo, (N, cin, H, W, cout, k, s, res, slope)))
print('cases %d, kernel runs %d, problems %d' % (ncase, nrun, len(bad)))
for b in bad[:30]: print('  ', b)
