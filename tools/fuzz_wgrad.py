"""Random weight-gradient shapes through yolo_conv_wgrad_algo (bf16 and fp32, every variant that accepts the shape, plus the
library's own choice) against torch's convolution weight gradient on the same rounded operands, computed on the GPU in fp32.
    python tools/fuzz_wgrad.py <seed> <seconds>"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from yolo_amd import lib as L
lib = L.load(); dev = torch.device('cuda:0')
torch.backends.cudnn.allow_tf32 = False
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
st = torch.cuda.current_stream().cuda_stream
ncase = nrun = 0
bad = []
t0 = time.time()
while time.time() - t0 < budget:
    k = int(rng.choice([1, 3])); s = int(rng.choice([1, 1, 2])) if k == 3 else 1
    cin = int(rng.choice([8, 16, 32, 64, 128, 192, 256, 384, 512]))
    cout = int(rng.choice([8, 32, 64, 90, 128, 192, 256, 512]))
    N = int(rng.choice([1, 2, 3, 8]))
    H = int(rng.integers(1, 60)); W = int(rng.integers(1, 70))
    if rng.random() < 0.25: H = W = int(rng.choice([13, 19, 26, 38, 52]))
    if N * H * W * max(cin, cout) > 8e6: continue
    dtype = 'f32' if rng.random() < 0.2 else 'bf16'
    ldt, tdt = (L.F32, torch.float32) if dtype == 'f32' else (L.BF16, torch.bfloat16)
    pad = k // 2; Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((N, H, W, cin), device=dev, generator=g).to(tdt)
    dy = torch.randn((N, Ho, Wo, cout), device=dev, generator=g).to(tdt)
    wt = torch.zeros((cout, cin, k, k), device=dev, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), wt, None, stride=s, padding=pad).backward(dy.float().permute(0, 3, 1, 2))
    ref = wt.grad
    scale = float(ref.abs().max()) + 1e-6
    ws = torch.zeros(max(int(lib.yolo_conv_wgrad_workspace_bytes(cin, cout, k, ldt)), 16), dtype=torch.uint8, device=dev)
    ncase += 1
    for algo in range(0, 8):
        dw = torch.ones((cout, cin, k, k), device=dev)
        rc = lib.yolo_conv_wgrad_algo(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, H, W, cin, cout, k, s, 0, ldt, ws.data_ptr(), algo, st)
        if rc != 0:
            if algo == 0 and (cout % 8 == 0 or dtype == 'f32'): bad.append(('algo 0 refused rc=%d' % rc, dtype, (N, cin, H, W, cout, k, s)))
            continue
        nrun += 1
        err = float(((dw - 1.0) - ref).abs().max()) / scale
        if not (err < (2e-3 if dtype == 'bf16' else 2e-4)):
            bad.append(('mismatch rel %.3g' % err, dtype, algo, (N, cin, H, W, cout, k, s)))
        if float(ws.view(torch.float32).abs().max() if ws.numel() >= 4 else 0) != 0.0:
            bad.append(('workspace not left zeroed', dtype, algo, (N, cin, H, W, cout, k, s)))
            ws.zero_()
print('cases %d, kernel runs %d, problems %d' % (ncase, nrun, len(bad)))
for b in bad[:30]: print('  ', b)
