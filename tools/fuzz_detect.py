"""Random detection configurations (image size, anchors per scale, class count, batch, logit scale, thresholds) through decode,
top-1 and per-class / objectness NMS against the oracle: rows to 1e-5, indices and kept ids exactly (on identical scores).
    python tools/fuzz_detect.py <seed> <seconds>"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo_amd.detect import Detector
from oracle import graph as og, detect as od
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
ncase, bad, t0 = 0, [], time.time()
base = og.spec_micro()
while time.time() - t0 < budget:
    A = int(rng.integers(1, 6)); ncls = int(rng.choice([1, 2, 4, 24, 40]))
    anchors = [[[float(rng.uniform(0.05, 0.9)), float(rng.uniform(0.05, 0.9))] for _ in range(A)] for _ in range(3)]
    spec = dict(base, slice_point=[1, 3, 5, 6, 6 + ncls], all_anchors=anchors)
    size = (32 * int(rng.integers(1, 7)), 32 * int(rng.integers(1, 7)))
    B = int(rng.choice([1, 2, 3]))
    steps = od.init_steps(spec['layers'], spec['all_anchors']); area = od.init_area(size, steps)
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])
    scale = float(rng.choice([0.0, 0.5, 2.0, 6.0]))
    outs = [(scale * rng.standard_normal((B, a, A, 6 + ncls))).astype(np.float32) for a in area]
    ctx = (size, A, ncls, B, scale)
    try:
        det = Detector(spec, size, steps, device=dev)
        devo = [torch.from_numpy(o).to(dev) for o in outs]
        with np.errstate(all='ignore'):
            rows = det.decode(devo).cpu().numpy()
            ref = od.decode_all(outs, spec['slice_point'], size, syxhw)
            if not np.allclose(rows, ref, rtol=1e-5, atol=1e-6, equal_nan=True): bad.append(('decode', float(np.nanmax(np.abs(rows - ref))), ctx))
            pred, idx = det.predict_device(devo)
            rpred, ridx = od.predict(outs, spec['slice_point'], size, syxhw)
            if idx.cpu().tolist() != ridx.tolist(): bad.append(('top-1 index', idx.cpu().tolist(), ridx.tolist(), ctx))
            # (y = (t + b) / 2 of a box whose half size dwarfs its centre -- exp(18) at logit scale 6 -- inherits one ulp of the
            #  exponential at the magnitude of the size: compared at that magnitude)
            elif not np.allclose(pred.cpu().numpy(), rpred, rtol=1e-5, atol=1e-6 + 4e-7 * float(np.nanmax(np.abs(np.where(np.isfinite(rpred), rpred, 0)))), equal_nan=True): bad.append(('top-1 row', ctx))
            for mode in ('class', 'obj'):
                r2, sc = det.decode_scores(devo, mode)
                kw = dict(valid_thresh=float(rng.choice([0.0, 0.01, 0.3])), iou_thresh=float(rng.choice([0.2, 0.45, 0.8])),
                          topk=int(rng.choice([1, 50, 400, 512])), post_nms=int(rng.choice([1, 10, 100])))
                for fast in (True, False):
                    kept, ks, cnt = det.nms(r2, mode, scores=sc, fast=fast, **kw)
                    for b in range(B):
                        rk, _ = od.nms(r2[b].cpu().numpy(), mode, scores=sc[b].cpu().numpy(), **kw)
                        if kept[b, :int(cnt[b])].cpu().tolist() != rk.tolist():
                            bad.append(('nms %s fast=%s' % (mode, fast), kw, ctx)); break
        ncase += 1
    except Exception as e:
        bad.append(('EXC', repr(e)[:200], ctx))
print('cases %d, problems %d' % (ncase, len(bad)))
for b in bad[:15]: print('  ', str(b)[:300])
