"""Random train-mode BatchNorm shapes (pixels x channels, fp32 and bf16, with and without the residual) through the three-launch
and the two-launch (_pp) forms of yolo_bn_train_fwd / _bwd against torch autograd on the GPU (fp32 math on the same rounded y).
    python tools/fuzz_bn.py <seed> <seconds>"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from yolo_amd import lib as L
lib = L.load(); dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
st = torch.cuda.current_stream().cuda_stream
ncase = 0
bad = []
t0 = time.time()
def close(name, a, b, rtol, atol, ctx):
    d = (a.float() - b.float()).abs()
    if not bool((d <= atol + rtol * b.float().abs()).all()):
        bad.append((name, 'max err %.3g' % float(d.max()), ctx))
while time.time() - t0 < budget:
    C_ = int(rng.choice([8, 16, 32, 64, 72, 128, 256, 512, 1000, 1024, 2048, 2112]))
    npix = int(rng.choice([1, 2, 3, 7, 64, 169, 1000, 4097, 21632, 86528]) if rng.random() < 0.5 else rng.integers(1, 30000))
    if npix * C_ > 3e7: continue
    dtype = 'f32' if rng.random() < 0.3 else 'bf16'
    ldt, tdt = (L.F32, torch.float32) if dtype == 'f32' else (L.BF16, torch.bfloat16)
    with_res = bool(rng.random() < 0.3)
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    y = (2 * torch.randn((npix, C_), device=dev, generator=g) + 0.5).to(tdt)
    res = torch.randn((npix, C_), device=dev, generator=g).to(tdt) if with_res else None
    dz = torch.randn((npix, C_), device=dev, generator=g).to(tdt)
    gamma = (0.5 + torch.rand(C_, device=dev, generator=g)); beta = 0.1 * torch.randn(C_, device=dev, generator=g)
    yf = y.float().requires_grad_(True); gam = gamma.clone().requires_grad_(True); bet = beta.clone().requires_grad_(True)
    mean = yf.mean(dim=0); var = yf.var(dim=0, unbiased=False)
    z = F.leaky_relu((yf - mean) / torch.sqrt(var + 1e-5) * gam + bet, 0.1)
    if with_res: z = z + res.float()
    z.backward(dz.float())
    ctx = (dtype, npix, C_, with_res)
    # LeakyReLU's kink: an element whose pre-activation is within rounding noise of 0 may take either slope, which moves ITS dy and
    # its channel's dgamma / dbeta (and through them every dy of the channel) by O(dz): such channels are compared on z only
    a_ref = ((yf - mean) / torch.sqrt(var + 1e-5) * gam + bet).detach()
    chan_ok = ~(a_ref.abs() < 2e-5 * (1 + a_ref.abs().amax(dim=0, keepdim=True))).any(dim=0)
    few = False                                          # (until round 3 the one-pass variance lost digits on a handful of nearly
                                                         #  equal values; the sums are now taken around a value of the channel)
    ncase += 1
    tol = (1e-4, 1e-5) if dtype == 'f32' else (1e-2, 1e-2)
    if npix < 16: tol = (tol[0] * 20, tol[1] * 20)            # (invstd up to 316: rounding noise of y - mean is amplified)
    for form in ('three', 'pp'):
        zd = torch.full_like(y, float('nan')); m_ = torch.empty(C_, device=dev); is_ = torch.empty(C_, device=dev)
        rm = torch.zeros(C_, device=dev); rv = torch.ones(C_, device=dev)
        dyd = torch.full_like(y, float('nan')); dg = torch.empty(C_, device=dev); db = torch.empty(C_, device=dev)
        if form == 'three':
            ws = torch.zeros(3 * C_, dtype=torch.float64, device=dev)
            rc = lib.yolo_bn_train_fwd(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), res.data_ptr() if with_res else None, zd.data_ptr(), m_.data_ptr(),
                                       is_.data_ptr(), rm.data_ptr(), rv.data_ptr(), ws.data_ptr(), npix, C_, 1e-5, 0.9, 0.1, ldt, st)
            rc2 = lib.yolo_bn_train_bwd(dz.data_ptr(), y.data_ptr(), m_.data_ptr(), is_.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dyd.data_ptr(),
                                        dg.data_ptr(), db.data_ptr(), ws.data_ptr(), npix, C_, 0.1, ldt, st) if rc == 0 else -9
        else:
            wa = torch.zeros(2 * 2112, dtype=torch.float64, device=dev); wb = torch.zeros(2 * 2112, dtype=torch.float64, device=dev)
            rc = lib.yolo_bn_train_fwd_pp(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), res.data_ptr() if with_res else None, zd.data_ptr(), m_.data_ptr(),
                                          is_.data_ptr(), rm.data_ptr(), rv.data_ptr(), wa.data_ptr(), wb.data_ptr(), 2 * 2112, npix, C_, 1e-5, 0.9, 0.1, ldt, st)
            rc2 = lib.yolo_bn_train_bwd_pp(dz.data_ptr(), y.data_ptr(), m_.data_ptr(), is_.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dyd.data_ptr(),
                                           dg.data_ptr(), db.data_ptr(), wb.data_ptr(), wa.data_ptr(), 2 * 2112, npix, C_, 0.1, ldt, st) if rc == 0 else -9
        if rc != 0 or rc2 != 0:
            bad.append(('refused', form, rc, rc2, ctx)); continue
        torch.cuda.synchronize()
        if torch.isnan(zd.float()).any() or torch.isnan(dyd.float()).any():
            bad.append(('NaN (unwritten) output', form, ctx)); continue
        if few:
            continue                                     # (see `few` above: launches and finite outputs only)
        close(form + ' z', zd, z.detach(), tol[0], tol[1], ctx)
        close(form + ' mean', m_, mean.detach(), 1e-4, 1e-5, ctx)
        close(form + ' running_var', rv, 0.9 + 0.1 * var.detach(), 1e-4, 1e-6, ctx)
        if not bool(chan_ok.any()):
            continue
        # (a batch of two pixels normalises to +-1: its true dy is ~0 and everything left is rounding noise of O(eps * dz) -- the
        #  scale of the comparison does not go below that; found by seed 61 in round 6)
        gscale = max(float(yf.grad[:, chan_ok].abs().max()), 1e-4 * float(dz.float().abs().max())) + 1e-6
        close(form + ' dy', dyd[:, chan_ok], yf.grad[:, chan_ok], tol[0] * 10, tol[1] * gscale * (1 if dtype == 'bf16' else 10), ctx)
        close(form + ' dgamma', dg[chan_ok], gam.grad[chan_ok], 2e-3, 2e-3 * (float(gam.grad[chan_ok].abs().max()) + 1e-6), ctx)
        close(form + ' dbeta', db[chan_ok], bet.grad[chan_ok], 2e-3, 2e-3 * (float(bet.grad[chan_ok].abs().max()) + 1e-6), ctx)
print('cases %d, problems %d' % (ncase, len(bad)))
for b in bad[:30]: print('  ', b)
