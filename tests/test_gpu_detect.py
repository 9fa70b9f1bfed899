"""GPU parity of decode / top-1 / IoU / NMS (csrc/detect.hip through the C ABI) against the oracle.
Index outputs must be bit-exact; float outputs within 1e-5 (libm exp differs by <=1 ulp)."""
import numpy as np
import pytest
import torch

from oracle import graph as og, detect as od

pytestmark = pytest.mark.gpu

SPEC = og.spec_d53()


def _setup(size, B, seed, cuda, scale=1.0):
    from yolo_amd.detect import Detector
    steps = od.init_steps(SPEC['layers'], SPEC['all_anchors'])
    area = od.init_area(size, steps)
    rng = np.random.default_rng(seed)
    outs = [(scale * rng.standard_normal((B, a, 3, 30))).astype(np.float32) for a in area]
    syxhw = od.init_syxhw(size, steps, SPEC['all_anchors'])
    det = Detector(SPEC, size, steps, device=cuda)
    return det, outs, syxhw, steps


@pytest.mark.parametrize('size', [(416, 416), (320, 512), (608, 608)])
def test_decode_and_top1(cuda, size):
    det, outs, syxhw, _ = _setup(size, 3, 11, cuda)
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows = det.decode(dev).cpu().numpy()
    ref = od.decode_all(outs, SPEC['slice_point'], size, syxhw)
    np.testing.assert_allclose(rows, ref, rtol=1e-5, atol=1e-6)
    pred, idx = det.predict_device(dev)
    rpred, ridx = od.predict(outs, SPEC['slice_point'], size, syxhw)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx)          # bit-exact index
    np.testing.assert_allclose(pred.cpu().numpy(), rpred, rtol=1e-5, atol=1e-6)
    assert det.predict(dev).dtype == np.float32


def test_top1_tie_lowest_index(cuda):
    """mxnet argmax returns the lowest index among ties."""
    det, outs, syxhw, _ = _setup((416, 416), 2, 12, cuda)
    for o in outs:
        o[..., 0] = -3.0
    outs[0][0, 100, 1, 0] = 2.5
    outs[1][0, 7, 2, 0] = 2.5            # same score, higher flat index
    outs[2][1, 5, 0, 0] = 1.0
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    _, idx = det.predict_device(dev)
    _, ridx = od.predict(outs, SPEC['slice_point'], (416, 416), syxhw)
    assert idx.cpu().tolist() == ridx.tolist() == [100 * 3 + 1, (2704 + 676 + 5) * 3]


def test_zero_logits_known_answer(cuda):
    """logits 0 -> score 0.5, centre = cell + stride/2, size = anchor."""
    det, outs, syxhw, steps = _setup((416, 416), 1, 13, cuda, scale=0.0)
    rows = det.decode([torch.from_numpy(o).to(cuda) for o in outs]).cpu().numpy()[0]
    assert np.all(rows[:, 0] == 0.5)
    k = (2704 + 5 * 26 + 7) * 3 + 1       # scale 1 (stride 16), cell (5,7), anchor 1
    l, t, r, b = rows[k, 1:5]
    ah, aw = SPEC['all_anchors'][1][1]
    np.testing.assert_allclose([(l + r) / 2, (t + b) / 2, r - l, b - t],
                               [(7 * 16 + 8) / 416., (5 * 16 + 8) / 416., aw, ah], rtol=1e-5)


def test_iou(cuda):
    from yolo_amd.detect import get_iou
    steps = od.init_steps(SPEC['layers'], SPEC['all_anchors'])
    ltrb = od.get_default_ltrb((416, 416), steps, SPEC['all_anchors'])
    target = np.asarray([3, 0.41, 0.52, 0.33, 0.27], np.float32)
    got = get_iou(torch.from_numpy(ltrb).to(cuda), torch.from_numpy(target), mode=2).cpu().numpy()
    ref = od.get_iou(ltrb, target, mode=2)
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7)
    assert int(np.argmax(got.reshape(-1))) == int(np.argmax(ref.reshape(-1)))
    # IoU(self) == 1, disjoint == 0
    box = np.asarray([[0.1, 0.2, 0.4, 0.6]], np.float32)
    t_self = np.asarray([0, 0.4, 0.25, 0.4, 0.3], np.float32)
    assert abs(float(get_iou(torch.from_numpy(box).to(cuda), torch.from_numpy(t_self), mode=2)[0, 0]) - 1.0) < 1e-6
    t_far = np.asarray([0, 0.9, 0.9, 0.05, 0.05], np.float32)
    assert float(get_iou(torch.from_numpy(box).to(cuda), torch.from_numpy(t_far), mode=2)[0, 0]) == 0.0


@pytest.mark.parametrize('mode', ['obj', 'class'])
@pytest.mark.parametrize('size', [(416, 416), (608, 608)])
def test_nms_indices_bit_exact(cuda, mode, size):
    det, outs, syxhw, _ = _setup(size, 2, 21, cuda, scale=2.0)
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows = det.decode(dev)
    scores = det.nms_scores(rows, mode)
    kept, ks, cnt = det.nms(rows, mode, scores=scores)
    rows_h, scores_h = rows.cpu().numpy(), scores.cpu().numpy()
    for b in range(rows_h.shape[0]):
        rk, rs = od.nms(rows_h[b], mode, scores=scores_h[b])
        n = int(cnt[b])
        assert n == len(rk)
        assert kept[b, :n].cpu().tolist() == rk.tolist()                      # bit-exact kept ids
        assert np.array_equal(ks[b, :n].cpu().numpy(), rs)
        assert (kept[b, n:] == -1).all()
    # scores themselves vs the oracle's own softmax / sigmoid
    ref_rows = od.decode_all(outs, SPEC['slice_point'], size, syxhw)
    _, rs0 = od.nms(ref_rows[0], mode)
    np.testing.assert_allclose(ks[0, :len(rs0)].cpu().numpy(), rs0, rtol=2e-5)


def test_nms_first_kept_is_top1(cuda):
    """The one invariant the reference pins: kept[0] == argmax(sigmoid(obj))."""
    det, outs, syxhw, _ = _setup((416, 416), 4, 22, cuda)
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows = det.decode(dev)
    kept, _, _ = det.nms(rows, 'obj')
    _, idx = det.predict_device(dev)
    assert kept[:, 0].cpu().tolist() == idx.cpu().tolist()


def test_nms_ties_and_few_candidates(cuda):
    """Equal scores resolve to the lower candidate id; fewer valid candidates than topk; empty image."""
    det, outs, syxhw, _ = _setup((416, 416), 3, 23, cuda)
    for o in outs:
        o[..., 0] = -20.0                      # sigmoid ~ 2e-9: below valid_thresh
    outs[0][0, 10:40, :, 0] = 1.25             # image 0: 90 boxes with identical objectness
    outs[0][1, 3, 0, 0] = 0.5                  # image 1: a single candidate
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows = det.decode(dev)
    scores = det.nms_scores(rows, 'obj')
    kept, ks, cnt = det.nms(rows, 'obj', scores=scores)
    rows_h, scores_h = rows.cpu().numpy(), scores.cpu().numpy()
    for b in range(3):
        rk, _ = od.nms(rows_h[b], 'obj', scores=scores_h[b])
        assert kept[b, :int(cnt[b])].cpu().tolist() == rk.tolist()
    assert int(cnt[1]) == 1 and int(cnt[2]) == 0
    # many exact ties across a topk boundary
    kept2, _, cnt2 = det.nms(rows, 'obj', topk=50, post_nms=50, iou_thresh=0.99, scores=scores)
    rk2, _ = od.nms(rows_h[0], 'obj', topk=50, post_nms=50, iou_thresh=0.99, scores=scores_h[0])
    assert kept2[0, :int(cnt2[0])].cpu().tolist() == rk2.tolist()


def test_cv_img_2_ndarray(cuda):
    from yolo_amd.detect import cv_img_2_ndarray
    img = np.random.default_rng(5).integers(0, 256, (246, 560, 3), dtype=np.uint8)   # licence_plate/test.jpg size
    got = cv_img_2_ndarray(img, cuda).cpu().numpy()
    np.testing.assert_array_equal(got, od.cv_img_2_ndarray(img))


def test_predict_LP_config1(cuda):
    """BASELINE config 1 plumbing: image -> (1,3,H,W)/255 on device, LPD pose decode of the best cell."""
    from yolo_amd.detect import predict_LP
    out = np.random.default_rng(6).standard_normal((1, 10, 10, 16)).astype(np.float32)
    out[0, 0, 3, 5] = out[0, 0, 7, 2] = 9.0                      # tie: the first cell wins
    got = predict_LP(torch.from_numpy(out).to(cuda), [40, 30, 20])
    ref, best = od.predict_LP(out, [40, 30, 20])
    assert best == 3 * 16 + 5
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('mode', ['obj', 'class'])
def test_nms_chip_wide_selection_equals_single_block(cuda, mode):
    """The workspace path (grid-wide radix passes + per-image list) and the single-block selection keep the same ids."""
    det, outs, syxhw, _ = _setup((608, 608), 5, 31, cuda, scale=2.0)
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows = det.decode(dev)
    scores = det.nms_scores(rows, mode)
    for kw in ({}, {'topk': 37, 'post_nms': 20}, {'valid_thresh': 0.3}, {'topk': 512, 'post_nms': 200, 'iou_thresh': 0.2}):
        a = det.nms(rows, mode, scores=scores, fast=True, **kw)
        b = det.nms(rows, mode, scores=scores, fast=False, **kw)
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_nms_list_overflow_falls_back(cuda):
    """All-equal scores (zero logits) put every candidate in one fine bucket: the list overflows and the
    single-block selection (ids ascending among ties) must take over on device."""
    det, outs, syxhw, _ = _setup((416, 416), 2, 32, cuda)
    for o in outs:
        o[...] = 0.0
    outs[0][1, 7, 1, 0] = 3.0                     # image 1: one candidate above the sea of ties
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows = det.decode(dev)
    for mode in ('obj', 'class'):
        scores = det.nms_scores(rows, mode)
        a = det.nms(rows, mode, scores=scores, fast=True)
        b = det.nms(rows, mode, scores=scores, fast=False)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        rk, _ = od.nms(rows[1].cpu().numpy(), mode, scores=scores[1].cpu().numpy())
        assert a[0][1, :int(a[2][1])].cpu().tolist() == rk.tolist()


def _extreme(kind, B=2):
    rng = np.random.default_rng(7)
    steps = od.init_steps(SPEC['layers'], SPEC['all_anchors'])
    area = od.init_area((416, 416), steps)
    o = [((40.0 if kind == 'huge' else 0.0 if kind == 'zeros' else 2.0) * rng.standard_normal((B, a, 3, 30))).astype(np.float32) for a in area]
    if kind == 'nan':
        o[0][0, 5, 1, 0] = np.nan              # an objectness: all 24 scores of that box are NaN
        o[1][1, 7, 2, 9] = np.nan              # one class logit: the box's softmax is NaN
    if kind == 'inf_obj':
        o[0][:, :, :, 0] = np.inf
    if kind == 'zero_area':
        o[2][:, :, :, 3:5] = -np.inf
    return o


@pytest.mark.parametrize('kind,kw', [('plain', dict(valid_thresh=0.999)), ('plain', dict(valid_thresh=0.0)), ('plain', dict(topk=512, post_nms=1)),
                                     ('plain', dict(topk=1)), ('plain', dict(iou_thresh=0.0)), ('plain', dict(iou_thresh=1.0)),
                                     ('zeros', {}), ('huge', {}), ('nan', {}), ('inf_obj', {}), ('zero_area', {})])
def test_nms_extremes_match_the_oracle(cuda, kind, kw):
    """Threshold / count extremes and non-finite logits: nothing valid, everything valid, one candidate, suppress-all and
    suppress-none IoU bars, mass ties, exp overflow (infinite boxes), NaN scores (never candidates), +inf scores (valid), zero-area
    boxes -- kept ids of the chip-wide and the single-block selection against the oracle's, image by image."""
    from yolo_amd.detect import Detector
    steps = od.init_steps(SPEC['layers'], SPEC['all_anchors'])
    det = Detector(SPEC, (416, 416), steps, device=cuda)
    outs = _extreme(kind)
    rows, scores = det.decode_scores([torch.from_numpy(o).to(cuda) for o in outs], 'class')
    with np.errstate(all='ignore'):
        want = [od.nms(rows[b].cpu().numpy(), 'class', scores=scores[b].cpu().numpy(), **kw)[0].tolist() for b in range(rows.shape[0])]
    for fast in (True, False):
        kept, _, cnt = det.nms(rows, 'class', scores=scores, fast=fast, **kw)
        for b in range(rows.shape[0]):
            assert kept[b, :int(cnt[b])].cpu().tolist() == want[b], (fast, b)
    if kind == 'plain' and kw.get('valid_thresh') == 0.999:
        assert cnt.cpu().tolist() == [0, 0]


@pytest.mark.parametrize('ncls', [1, 2, 24, 58, 90])
def test_decode_scores_any_class_count(cuda, ncls):
    """The pipelined decode + scores kernel for row widths other than the car spec's 30: one class (C = 7), an odd tile tail, the
    24-load-per-thread instantiation (C > 32) and its 96-value limit; bit-identical to yolo_decode + yolo_nms_scores, and the
    scores are the oracle's softmax x objectness."""
    from yolo_amd.detect import Detector
    spec = dict(SPEC, slice_point=[1, 3, 5, 6, 6 + ncls])
    size, B = (160, 224), 3
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    rng = np.random.default_rng(100 + ncls)
    outs = [(2.0 * rng.standard_normal((B, a, 3, 6 + ncls))).astype(np.float32) for a in area]
    det = Detector(spec, size, steps, device=cuda)
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows = det.decode(dev)
    for mode in ('obj', 'class'):
        scores = det.nms_scores(rows, mode)
        rows2, scores2 = det.decode_scores(dev, mode)
        assert torch.equal(rows.view(torch.int32), rows2.view(torch.int32))
        assert torch.equal(scores.view(torch.int32), scores2.view(torch.int32))
    r = rows.cpu().numpy()
    logits = r[..., 6:]
    e = np.exp(logits - logits.max(axis=-1, keepdims=True))
    want = r[..., 0:1] * (e / e.sum(axis=-1, keepdims=True))
    np.testing.assert_allclose(det.nms_scores(rows, 'class').cpu().numpy().reshape(want.shape), want, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('mode', ['obj', 'class'])
@pytest.mark.parametrize('size,B', [((416, 416), 3), ((608, 608), 2), ((320, 512), 1)])
def test_decode_scores_fused_is_bit_identical(cuda, mode, size, B):
    """yolo_decode_scores (one pass over the logits) against yolo_decode + yolo_nms_scores: identical bits, including a
    last block that holds fewer than 256 boxes."""
    det, outs, syxhw, _ = _setup(size, B, 41, cuda, scale=2.0)
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows = det.decode(dev)
    scores = det.nms_scores(rows, mode)
    rows2, scores2 = det.decode_scores(dev, mode)
    assert torch.equal(rows.view(torch.int32), rows2.view(torch.int32))
    assert torch.equal(scores.view(torch.int32), scores2.view(torch.int32))
    a = det.nms(rows, mode, scores=scores)
    b = det.nms(rows2, mode, scores=scores2)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize('mode', ['obj', 'class'])
@pytest.mark.parametrize('size,B,kw', [((416, 416), 3, {}), ((608, 608), 2, {}), ((320, 512), 5, dict(valid_thresh=0.0)),
                                       ((160, 224), 7, dict(valid_thresh=0.3, topk=50)), ((416, 416), 2, dict(valid_thresh=0.999))])
def test_decode_nms_one_call_equals_the_two_calls(cuda, mode, size, B, kw):
    """yolo_decode_nms (the decode pass takes the selection's first histogram from the scores it holds in LDS) against
    yolo_decode_scores + yolo_nms_from_scores: identical rows, scores, kept ids / scores / counts -- also where a 128-box tile
    straddles two images (10 647 or 22 743 boxes per image are not multiples of 128), with every score valid and
    with none -- and against the oracle's greedy NMS on the same scores."""
    det, outs, syxhw, _ = _setup(size, B, 77, cuda, scale=2.0)
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    rows, scores = det.decode_scores(dev, mode)
    a = det.nms(rows, mode, scores=scores, **kw)
    rows2, scores2, kept, ks, cnt = det.decode_nms(dev, mode, **kw)
    assert torch.equal(rows.view(torch.int32), rows2.view(torch.int32))
    assert torch.equal(scores.view(torch.int32), scores2.view(torch.int32))
    for x, y in zip(a, (kept, ks, cnt)):
        assert torch.equal(x, y)
    rows_h, scores_h = rows.cpu().numpy(), scores.cpu().numpy()
    for b in range(B):
        rk, _ = od.nms(rows_h[b], mode, scores=scores_h[b], **kw)
        assert kept[b, :int(cnt[b])].cpu().tolist() == rk.tolist()
    # the histogram the decode pass took IS the selection pass's: counts of valid scores by their top 11 bits, per image
    ws = det._nms_ws[B]
    hist = ws[:B * 2 * 2048 * 4].view(torch.int32).view(B, 2, 2048)[:, 0].cpu().numpy()
    vt = np.float32(max(kw.get('valid_thresh', 0.01), 0.0))
    for b in range(B):
        u = scores_h[b].view(np.uint32)
        ok = (u >= vt.view(np.uint32)) & (u <= 0x7f800000)
        want = np.bincount((u[ok] >> 21).astype(np.int64), minlength=2048)
        assert np.array_equal(hist[b], want), b


def test_decode_nms_rejects(cuda):
    from yolo_amd import lib as L
    import ctypes as C
    det, outs, _, _ = _setup((416, 416), 1, 3, cuda)
    lib = L.load()
    p = torch.zeros(1 << 16, device=cuda).data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    f = C.c_float
    assert lib.yolo_decode_nms(p, p, p, 1, 30, C.byref(det.grid), 1, f(0.01), f(0.45), 400, 100, p, p, p, None, st) == -1      # no workspace
    assert lib.yolo_decode_nms(p, p, p, 1, 30, C.byref(det.grid), 1, f(0.01), f(0.45), 600, 100, p, p, p, p, st) == -2       # top-k beyond the sort
    assert lib.yolo_decode_nms(p, p, p, 0, 30, C.byref(det.grid), 1, f(0.01), f(0.45), 400, 100, p, p, p, p, st) == -1
    assert lib.yolo_decode_nms(p, p, p, 1, 30, None, 1, f(0.01), f(0.45), 400, 100, p, p, p, p, st) == -1


def test_predict_async_equals_predict(cuda):
    """Detector.predict_async (the rows copied into a pinned host buffer behind the kernels, two slots) returns predict()'s rows;
    a slot's buffer is stable until the slot is used again, and alternating slots keeps two frames apart."""
    det, outs, syxhw, _ = _setup((416, 416), 3, 51, cuda, scale=2.0)
    dev = [torch.from_numpy(o).to(cuda) for o in outs]
    want = det.predict(dev)
    h0, e0 = det.predict_async(dev, slot=0)
    dev2 = [d * 0.5 for d in dev]
    want2 = det.predict(dev2)
    h1, e1 = det.predict_async(dev2, slot=1)
    e0.synchronize(); e1.synchronize()
    assert h0.dtype == np.float32 and h0.shape == want.shape
    np.testing.assert_array_equal(h0, want)
    np.testing.assert_array_equal(h1, want2)
    assert not np.array_equal(h0, h1)
    h0b, e0b = det.predict_async(dev2, slot=0)              # the slot's buffer is re-used
    e0b.synchronize()
    assert h0b.ctypes.data == h0.ctypes.data
    np.testing.assert_array_equal(h0b, want2)
