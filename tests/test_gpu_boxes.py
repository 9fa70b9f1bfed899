"""The `north_star` tolerance -- "outputs (box xywh, class scores, kept indices) match the reference within 1e-3" -- stated on the
DECODED outputs of every arithmetic path, at the BASELINE configurations (Missing 3: only the logits' RMS was
bounded for the reduced-precision path).  For configs[1] (D53 spec 416x416 bs 32, images 0 / 1 / 31 of the batch) and configs[4]'s
shape (608x608, two images) the HIP net runs in dtype f32 / f16 / bf16, `Detector.decode` + `predict` turn the logits into the
reference's rows (car/YOLO.py:552-597), and the rows are compared with the fp32 oracle's: max and RMS of |a - b| / (1 + |b|) over [l, t, r, b]
(normalised image units, as the reference emits them), of the top-1 `predict` row [score, y, x, h, w], and the fraction of images
whose top-1 box index equals the oracle's.

What is BARRED: fp32 <= 1e-3 everywhere (the parity path).  What is STATED (printed, written to gpurun_out/box_error.json, quoted in
DESIGN.md 5): the reduced-precision paths' numbers -- with loose sanity bars so that a broken path cannot hide behind "stated"."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import graph as og, forward as of, detect as od
from util import KERNEL_SETS, pin_kernels, assert_plan_held

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# sanity bars of the stated numbers (not parity bars): RMS error of the box coordinates over ALL boxes, image units
SANITY_RMS = {'f32': 1e-4, 'bf16x3': 1e-4, 'f16x3': 1e-4, 'f16': 2e-3, 'bf16': 2e-2}
# the paths held to the north-star bar (1e-3 on every decoded box, score and top-1 row; the same top-1 box): exact fp32 and split bf16
BARRED = ('f32', 'bf16x3', 'f16x3')


def _stats(rows, ref_rows, pred, ref_pred, idx, ref_idx):
    # error measure: |a - b| / (1 + |b|) -- absolute for boxes of image size (the reference's rows are normalised 0..1), relative
    # for the huge ones a random-weight net also emits (exp(th) * anchor with th ~ 10: an absolute bar is meaningless there).
    # The same measure tests/test_gpu_configs.py::test_d53_608_forward holds the fp32 path to.
    e = (rows[..., 1:5].astype(np.float64) - ref_rows[..., 1:5]) / (1.0 + np.abs(ref_rows[..., 1:5]))
    ep = (pred[:, :5].astype(np.float64) - ref_pred[:, :5]) / (1.0 + np.abs(ref_pred[:, :5]))
    same = idx == ref_idx
    # the top-1 row compared where both paths picked the same box (another box is another object, not a rounding error)
    eps = ep[same] if same.any() else np.zeros((1, 5))
    return {'box_ltrb_max': float(np.abs(e).max()), 'box_ltrb_rms': float(np.sqrt(np.mean(e * e))),
            'box_ltrb_p999': float(np.quantile(np.abs(e), 0.999)),
            'score_max': float(np.abs(rows[..., 0].astype(np.float64) - ref_rows[..., 0]).max()),
            'predict_row_max': float(np.abs(eps).max()), 'predict_row_rms': float(np.sqrt(np.mean(eps * eps))),
            'top1_index_agreement': float(np.mean(same)), 'images': int(len(idx)), 'boxes_per_image': int(rows.shape[1])}


def _record(key, st):
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, 'box_error.json')
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = st
    with open(path, 'w') as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print('box error %s: %s' % (key, json.dumps(st)))


def _run(cuda, dtype, size, B, sel, seed, kernels='measured', rep=1):
    """rep > 1: the B images repeated rep times (a, b, a, b, ...) -- the committed plan holds the shapes of 608x608 at bs 64, the
    oracle cannot hold 64 different images of that size; eval-mode images are independent."""
    from yolo_amd.net import CarNet
    from yolo_amd.detect import Detector
    spec = og.spec_d53()
    g = og.build_graph(spec)
    P = og.init_params(g, seed=0, bn='random')
    x = np.random.default_rng(seed).random((B, 3) + size, dtype=np.float32)
    net = CarNet(spec, dtype=dtype, device=cuda, tune='measure').load_params(P)
    pinned = pin_kernels(net, kernels)
    outs = net(torch.from_numpy(x).to(cuda).repeat(rep, 1, 1, 1))
    assert_plan_held(net, pinned, 'boxes_%dx%d_bs%d_%s' % (size[0], size[1], B * rep, dtype))
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])
    det = Detector(spec, size, steps, device=cuda)
    rows = det.decode(outs).cpu().numpy()[sel]
    pred, idx = det.predict_device(outs)
    pred, idx = pred.cpu().numpy()[sel], idx.cpu().numpy()[sel]
    # per-class NMS end to end: the HIP net's logits through the HIP decode + NMS against the ORACLE's logits through the oracle's
    # decode + NMS (north_star: "kept indices").  The ids are exact on identical inputs (tests/test_gpu_detect.py); across arithmetic
    # paths a pair of candidates whose scores differ by less than the path's error may swap ranks: stated as the fraction of images
    # whose kept list is identical and the mean overlap of the kept sets
    rows_d = det.decode(outs)
    kept, _, cnt = det.nms(rows_d, 'class', scores=det.nms_scores(rows_d, 'class'))
    kept, cnt = kept.cpu().numpy()[sel], cnt.cpu().numpy()[sel]
    ref = [r.numpy() for r in of.forward_torch(g, P, x[sel])]
    ref_rows = od.decode_all(ref, spec['slice_point'], size, syxhw)
    ref_pred, ref_idx = od.predict(ref, spec['slice_point'], size, syxhw)
    st = _stats(rows, ref_rows, pred, ref_pred, idx, ref_idx)
    same, jac = [], []
    for i in range(len(sel)):
        rk, _ = od.nms(ref_rows[i], mode='class')
        got = [int(v) for v in kept[i, :int(cnt[i])]]
        same.append(got == [int(v) for v in rk])
        jac.append(len(set(got) & set(rk)) / float(max(1, len(set(got) | set(rk)))))
    st['nms_kept_lists_identical'] = float(np.mean(same))
    st['nms_kept_sets_overlap'] = float(np.mean(jac))
    return st


@pytest.mark.parametrize('kernels', KERNEL_SETS)
@pytest.mark.parametrize('dtype', ['f32', 'f16x3', 'bf16x3', 'f16', 'bf16'])
def test_config1_box_error_vs_fp32_oracle(cuda, dtype, kernels):
    st = _run(cuda, dtype, (416, 416), 32, [0, 1, 31], seed=7, kernels=kernels)
    _record('configs1_416_bs32_%s%s' % (dtype, '' if kernels == 'measured' else '_plan'), st)
    assert np.isfinite(list(v for v in st.values())).all()
    assert st['box_ltrb_rms'] < SANITY_RMS[dtype], st
    if dtype in BARRED:
        assert st['box_ltrb_max'] <= 1e-3 and st['score_max'] <= 1e-3 and st['predict_row_max'] <= 1e-3 and st['top1_index_agreement'] == 1.0, st
        assert st['nms_kept_sets_overlap'] >= 0.98, st


@pytest.mark.parametrize('kernels', KERNEL_SETS)
@pytest.mark.parametrize('dtype', ['f32', 'f16x3', 'bf16x3', 'f16', 'bf16'])
def test_config4_box_error_vs_fp32_oracle(cuda, dtype, kernels):
    if kernels == 'plan' and dtype == 'f32':
        pytest.skip('the committed plan holds no fp32 shapes at 608x608 (bench.py runs f32 at 416x416 bs 32 only)')
    # ('plan': the plan's shapes are bs 64 -- the two images 32 times over)
    st = _run(cuda, dtype, (608, 608), 2, [0, 1], seed=5, kernels=kernels, rep=32 if kernels == 'plan' else 1)
    _record('configs4_608_%s%s' % (dtype, '' if kernels == 'measured' else '_plan'), st)
    assert st['box_ltrb_rms'] < SANITY_RMS[dtype], st
    if dtype in BARRED:
        assert st['box_ltrb_max'] <= 1e-3 and st['score_max'] <= 1e-3 and st['predict_row_max'] <= 1e-3 and st['top1_index_agreement'] == 1.0, st
