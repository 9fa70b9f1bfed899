"""Random network specs (stage depths and widths in the reference's schema), image sizes (multiples of the total stride), batches
and fusion switches through CarNet.forward against the oracle: fp32 logits to 1e-3, bf16 within 1.5x the rounding-aware oracle's
own distance to fp32 (the bars of DESIGN 5); the fused and the unfused build of the same net must agree like two bf16 evaluations.    python tools/fuzz_net.py <seed> <seconds>"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo_amd.net import CarNet
from oracle import graph as og, forward as of
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
ncase, bad, t0 = 0, [], time.time()
while time.time() - t0 < budget:
    base = int(rng.choice([8, 16, 32]))
    nst = 5
    layers = [1] + [int(rng.integers(1, 4)) for _ in range(nst - 1)]
    channels = [base * (2 ** i) for i in range(nst + 1)]
    if rng.random() < 0.3: channels = [32, 64, 128, 128, 256, 256]          # (the fused stem / residual kernels need 32 -> 64 -> 128)
    ncls = int(rng.choice([1, 4, 24]))
    spec = dict(layers=layers, channels=channels, slice_point=[1, 3, 5, 6, 6 + ncls], all_anchors=og.CAR_ANCHORS)
    H, W = 32 * int(rng.integers(1, 8)), 32 * int(rng.integers(1, 8))
    B = int(rng.choice([1, 2, 3, 5]))
    if B * H * W * channels[1] > 4e7: continue
    u_ = rng.random()
    dtype = 'f32' if u_ < 0.15 else ('bf16x3' if u_ < 0.4 else 'f16x3' if u_ < 0.65 else 'bf16')      # (split types, round 6: held to the fp32 bar)
    try:
        g = og.build_graph(spec)
        P = og.init_params(g, seed=int(rng.integers(1000)), bn='random')
        x = rng.random((B, 3, H, W), dtype=np.float32)
        kw = dict(fuse_stem=bool(rng.random() < 0.7), fuse_res=bool(rng.random() < 0.7), fuse_concat=bool(rng.random() < 0.7), side_stream=bool(rng.random() < 0.5))
        if dtype in ('bf16x3', 'f16x3'):
            kw['fuse_concat'] = True                 # (the split type has no up-sample + concat copy kernel: the fused form only)
        net = CarNet(spec, dtype=dtype, device=dev, **kw).load_params(P)
        outs = [o.cpu().numpy() for o in net(torch.from_numpy(x).to(dev))]
        ref = [r.numpy() for r in of.forward_torch(g, P, x)]
        ctx = (dtype, layers, channels, ncls, (H, W), B, kw)
        ncase += 1
        if dtype in ('f32', 'bf16x3', 'f16x3'):
            e = max(float(np.abs(o - r).max()) for o, r in zip(outs, ref))
            if not e < 1e-3: bad.append(('%s logits' % dtype, e, ctx))
        else:
            sim = [r.numpy() for r in of.forward_torch(g, P, x, sim_bf16=True)]
            rms = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
            num = sum(rms(o, r) ** 2 for o, r in zip(outs, ref)) ** 0.5; den = sum(rms(s_, r) ** 2 for s_, r in zip(sim, ref)) ** 0.5
            scale = sum(float(np.mean(r ** 2)) for r in ref) ** 0.5
            if not (num <= 1.5 * den + 1e-4 * scale or num < 0.015 * scale): bad.append(('bf16 distance to fp32 %.3g vs simulated %.3g (scale %.3g)' % (num, den, scale), ctx))
            plain = CarNet(spec, dtype=dtype, device=dev, fuse_stem=False, fuse_res=False, fuse_concat=False, side_stream=False).load_params(P)
            outs2 = [o.cpu().numpy() for o in plain(torch.from_numpy(x).to(dev))]
            # (a conv that writes into its half of a concat buffer runs the pipelined kernel where the dense one may run the streaming
            #  kernel: chunk-major against tap-major K order, i.e. a different fp32 summation order -- last-bit differences that a
            #  bf16 rounding occasionally turns into one ulp; the builds must agree like two bf16 evaluations, not bit for bit)
            dmax = max(float(np.abs(a - b).max()) for a, b in zip(outs, outs2))
            if not dmax < 0.03 * max(float(np.abs(r).max()) for r in ref):
                bad.append(('fused and unfused builds differ', dmax, ctx))
    except Exception as e:
        bad.append(('EXC', repr(e)[:200], (dtype, layers, channels, ncls, (H, W), B)))
print('cases %d, problems %d' % (ncase, len(bad)))
for b in bad[:12]: print('  ', str(b)[:400])
