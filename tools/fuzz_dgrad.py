"""Random data-gradient shapes: the stride-1 / dilated form (yolo_pack_conv_weights_dgrad + yolo_conv_fwd, what the Trainer launches)
and the sub-pixel stride-2 form (yolo_pack_conv_weights_dgrad_s2 + yolo_conv_dgrad_s2), every variant that accepts the shape,
against torch autograd on the GPU (fp32 math on the bf16-rounded operands).
    PYTORCH_NO_CUDA_MEMORY_CACHING=1 python tools/fuzz_dgrad.py <seed> <seconds>
(without the variable torch's OWN convolution-backward kernels ran into a memory access fault on one shape sequence here: every
tensor its own allocation also turns out-of-bounds reads of the kernels under test into faults instead of hiding them)"""
import sys, os, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
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
    k = int(rng.choice([1, 3])); s = int(rng.choice([1, 2])) if k == 3 else 1
    cin = int(rng.choice([8, 16, 32, 64, 128, 256, 512])); cout = int(rng.choice([8, 32, 64, 128, 256, 512]))
    N = int(rng.choice([1, 2, 3, 6]))
    H = int(rng.integers(1, 30)) * s; W = int(rng.integers(1, 36)) * s
    if os.environ.get('FUZZ_S2_REGULAR'):                 # large regular maps only (dy width a multiple of 8, >= 1024 pixels)
        k, s = 3, 2; W = 16 * int(rng.integers(1, 14)); lo = max(1, 2048 // W); H = 2 * int(rng.integers(lo, lo + 40))
        cin = int(rng.choice([8, 16, 32, 64, 128])); cout = int(rng.choice([32, 64, 128, 256])); N = int(rng.choice([1, 2, 3, 5]))
    if N * H * W * max(cin, cout) > (3e7 if os.environ.get('FUZZ_S2_REGULAR') else 6e6): continue
    pad = k // 2; Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    w = torch.randn((cout, cin, k, k), device=dev, generator=g) / float(np.sqrt(cin * k * k))
    dy = torch.randn((N, Ho, Wo, cout), device=dev, generator=g).to(torch.bfloat16)
    xt = torch.zeros((N, cin, H, W), device=dev, requires_grad=True)
    F.conv2d(xt, w.to(torch.bfloat16).float(), None, stride=s, padding=pad).backward(dy.float().permute(0, 3, 1, 2))
    ref = xt.grad.permute(0, 2, 3, 1).contiguous()
    scale = float(ref.abs().max()) + 1e-6
    ctx = (N, cin, H, W, cout, k, s)
    ncase += 1
    if os.environ.get('FUZZ_TRACE'): torch.cuda.synchronize(); print('case', ctx, '(torch reference done)', flush=True)
    def check(got, name, algo):
        global nrun
        nrun += 1
        if torch.isnan(got.float()).any():
            bad.append((name, 'NaN (unwritten)', algo, ctx)); return
        e = float((got.float() - ref).abs().max()) / scale
        if not e < 1.5e-2:
            bad.append((name, 'rel err %.3g' % e, algo, ctx))
    if s == 2:
        wd = torch.empty(max(int(lib.yolo_packed_weight_bytes(4 * cin, cout, 2, L.BF16)), 16), dtype=torch.uint8, device=dev)
        rc = lib.yolo_pack_conv_weights_dgrad_s2(w.data_ptr(), wd.data_ptr(), cout, cin, L.BF16, st)
        if rc == 0:
            cp = lib.yolo_padded_channels(4 * cin)
            ones = torch.ones(cp, device=dev); zeros = torch.zeros(cp, device=dev)
            for algo in (0, 2, 6, 10, 4):
                out = torch.full((N, H, W, cin), float('nan'), dtype=torch.bfloat16, device=dev)
                d = L.ConvDesc()
                d.x, d.w_packed, d.scale, d.bias, d.y = dy.data_ptr(), wd.data_ptr(), ones.data_ptr(), zeros.data_ptr(), out.data_ptr()
                d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = N, Ho, Wo, cout, 4 * cin, 2, 1, L.BF16, 1.0, algo
                if os.environ.get('FUZZ_TRACE'): print('  s2 algo', algo, flush=True)
                if lib.yolo_conv_dgrad_s2(C.byref(d), st) == 0:
                    if os.environ.get('FUZZ_TRACE'): torch.cuda.synchronize()
                    check(out, 'sub-pixel s2', algo)
    # the stride-1 form (stride 2: over the dilated gradient, as the Trainer's fallback does)
    wdg = torch.zeros(max(int(lib.yolo_packed_weight_bytes(cin, cout, k, L.BF16)), 16), dtype=torch.uint8, device=dev)
    if lib.yolo_pack_conv_weights_dgrad(w.data_ptr(), wdg.data_ptr(), cout, cin, k, L.BF16, st) != 0:
        continue
    src, sshape = dy, (N, Ho, Wo, cout)
    if s == 2:
        dil = torch.empty((N, H, W, cout), dtype=torch.bfloat16, device=dev)
        if lib.yolo_dilate2x(dy.data_ptr(), dil.data_ptr(), N, H, W, Ho, Wo, cout, L.BF16, st) != 0:
            bad.append(('dilate2x refused', ctx)); continue
        src, sshape = dil, (N, H, W, cout)
    for algo in (0,) + tuple(CarNet.ALGOS):
        out = torch.full((N, H, W, cin), float('nan'), dtype=torch.bfloat16, device=dev)
        d = L.ConvDesc()
        d.x, d.w_packed, d.y = src.data_ptr(), wdg.data_ptr(), out.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = sshape[0], sshape[1], sshape[2], cout, cin, k, 1, L.BF16, 1.0, algo
        if os.environ.get('FUZZ_TRACE'): print('  s1 algo', algo, flush=True)
        rc = lib.yolo_conv_fwd(C.byref(d), st)
        if os.environ.get('FUZZ_TRACE'): torch.cuda.synchronize()
        if rc != 0:
            if algo == 0: bad.append(('stride-1 form: algo 0 refused rc=%d' % rc, ctx))
            continue
        check(out, 'stride-1 form', algo)
print('cases %d, kernel runs %d, problems %d' % (ncase, nrun, len(bad)))
for b in bad[:25]: print('  ', b)
