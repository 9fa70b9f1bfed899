"""The oracle's decode / top-1 / IoU against a second, independent statement of the same arithmetic that the
reference holds (its insulator detector: insulator/YOLO.py:306-341, insulator/utils.py:65-98),
restated in oracle/witness_insulator.py -- on CPU between the two restatements, and (gpu) against the HIP kernels.
Plus BASELINE configs[0]: licence_plate/test.jpg through the whole path."""
import os

import numpy as np
import pytest

from oracle import detect as od, graph as og, witness_insulator as wi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SPEC = og.spec_d53()


def _case(size, seed, B=1, spread=2.0):
    steps = od.init_steps(SPEC['layers'], SPEC['all_anchors'])
    n = sum(od.init_area(size, steps))
    out = (spread * np.random.default_rng(seed).standard_normal((B, n, 3, 30))).astype(np.float32)
    return steps, out


@pytest.mark.parametrize('size', [(416, 416), (320, 512), (608, 608)])
def test_grid_and_decode_agree_with_the_second_witness(size):
    steps, out = _case(size, 11)
    consts = wi.grid(size, steps, SPEC['all_anchors'])
    for a, b in zip(consts, od.init_syxhw(size, steps, SPEC['all_anchors'])):
        np.testing.assert_array_equal(a, np.asarray(b, np.float32).reshape(a.shape))
    ltrb_w = wi.yxhw_to_ltrb(out[..., 1:5], size, consts)
    ltrb_o = od.yxhw_to_ltrb(out[..., 1:5], size, od.init_syxhw(size, steps, SPEC['all_anchors']))
    np.testing.assert_array_equal(ltrb_w, np.asarray(ltrb_o, np.float32))           # same fp32 operations: bit-equal


def test_predict_agrees_with_the_second_witness():
    size = (416, 416)
    steps, out = _case(size, 12, B=3)
    syxhw = od.init_syxhw(size, steps, SPEC['all_anchors'])
    pred, idx = od.predict([out], SPEC['slice_point'], size, syxhw)
    consts = wi.grid(size, steps, SPEC['all_anchors'])
    for b in range(3):
        row, best = wi.predict(out[b:b + 1, ..., 0:1], out[b:b + 1, ..., 1:5], out[b:b + 1, ..., 5:], size, consts)
        assert best == int(idx[b])
        np.testing.assert_array_equal(row, pred[b])
    # ties: the first index wins in both
    out[0, :, :, 0] = 0.25
    row, best = wi.predict(out[:1, ..., 0:1], out[:1, ..., 1:5], out[:1, ..., 5:], size, consts)
    pred, idx = od.predict([out[:1]], SPEC['slice_point'], size, syxhw)
    assert best == int(idx[0]) == 0


@pytest.mark.parametrize('mode', [1, 2])
def test_iou_agrees_with_the_second_witness(mode):
    rng = np.random.default_rng(13)
    c = rng.random((500, 3, 2)).astype(np.float32)
    wh = (rng.random((500, 3, 2)) * 0.4).astype(np.float32)
    ltrb = np.concatenate([c - wh / 2, c + wh / 2], axis=-1).astype(np.float32)
    target = np.asarray([3, 0.45, 0.55, 0.3, 0.2], np.float32) if mode == 2 else np.asarray([3, 0.3, 0.25, 0.7, 0.8], np.float32)
    a, b = wi.get_iou(ltrb, target, mode), od.get_iou(ltrb, target, mode)
    assert a.shape == (500, 3, 1)
    np.testing.assert_array_equal(a, b)
    # hand-computed known answers.  mode 2: box == target -> 1; disjoint -> 0; half overlap of equal boxes -> 1/3
    t2 = np.asarray([0, 0.5, 0.5, 0.2, 0.4], np.float32)                    # y, x, h, w -> l .3 t .4 r .7 b .6
    boxes = np.asarray([[0.3, 0.4, 0.7, 0.6], [0.8, 0.8, 0.9, 0.9], [0.5, 0.4, 0.9, 0.6]], np.float32)
    np.testing.assert_allclose(wi.get_iou(boxes, t2, 2).ravel(), [1.0, 0.0, 1.0 / 3.0], rtol=1e-5)
    # mode 1 keeps the reference's target_area = target[3] * target[4] = r * b (insulator/utils.py:96, yolo_gluon.py:166):
    # target l .3 t .4 r .7 b .6 against itself: inter .08, "target area" .7 * .6 = .42 -> .08 / (.08 + .42 - .08)
    t1 = np.asarray([0, 0.3, 0.4, 0.7, 0.6], np.float32)
    np.testing.assert_allclose(wi.get_iou(boxes[:1], t1, 1).ravel(), [0.08 / 0.42], rtol=1e-5)
    np.testing.assert_allclose(od.get_iou(boxes[:1], t1, 1).ravel(), [0.08 / 0.42], rtol=1e-5)


@pytest.mark.gpu
def test_hip_decode_top1_iou_against_the_second_witness(cuda):
    import torch
    from yolo_amd.detect import Detector, get_iou
    size = (608, 608)
    steps, out = _case(size, 14, B=2)
    consts = wi.grid(size, steps, SPEC['all_anchors'])
    det = Detector(SPEC, size, steps, device=cuda)
    t = torch.from_numpy(out).to(cuda)
    rows = det.decode(t).cpu().numpy().reshape(2, -1, 3, 30)
    ref = wi.yxhw_to_ltrb(out[..., 1:5], size, consts)
    np.testing.assert_allclose(rows[..., 1:5], ref, rtol=1e-5, atol=1e-6)
    pred, idx = det.predict_device(t)
    for b in range(2):
        row, best = wi.predict(out[b:b + 1, ..., 0:1], out[b:b + 1, ..., 1:5], out[b:b + 1, ..., 5:], size, consts)
        assert int(idx[b]) == best                                                       # bit-exact index
        np.testing.assert_allclose(pred[b].cpu().numpy(), row, rtol=1e-5, atol=1e-6)
    ltrb = torch.from_numpy(ref[0]).to(cuda)
    for mode, target in ((1, [3, 0.3, 0.25, 0.7, 0.8]), (2, [3, 0.45, 0.55, 0.3, 0.2])):
        tg = np.asarray(target, np.float32)
        got = get_iou(ltrb, torch.from_numpy(tg), mode=mode).cpu().numpy()
        np.testing.assert_allclose(got, wi.get_iou(ref[0], tg, mode), rtol=1e-5, atol=1e-7)
    # the reference's default is mode 1 (yolo_gluon.py:127): a caller that omits `mode` gets it
    np.testing.assert_array_equal(get_iou(ltrb, torch.from_numpy(tg)).cpu().numpy(), get_iou(ltrb, torch.from_numpy(tg), mode=1).cpu().numpy())
    with pytest.raises(ValueError):
        get_iou(ltrb, torch.from_numpy(tg), mode=3)


# ---- BASELINE configs[0]: licence_plate/test.jpg, 416x416, batch 1 (plumbing) ----------------------------------------
def _test_image():
    """tests/golden/lp_test.jpg = the reference's licence_plate/test.jpg (560x246 RGB, a data file).  Decoded and resized
    to (W=416, H=416) bilinear as `cv2.resize(img, (w, h))` does in the reference's LPD node (LPD_video_node.py) -- with
    PIL, the decoder available here; both sides of every comparison below see the same array."""
    from PIL import Image
    im = Image.open(os.path.join(GOLD, 'lp_test.jpg')).convert('RGB')
    assert im.size == (560, 246)
    return np.asarray(im.resize((416, 416), Image.BILINEAR), np.uint8)


def test_config0_plumbing_on_cpu():
    """The oracle's side of config 0: image -> cv_img_2_ndarray -> shapes, dtype, range, determinism."""
    img = _test_image()
    assert img.shape == (416, 416, 3) and img.dtype == np.uint8
    x = od.cv_img_2_ndarray(img)
    assert x.shape == (1, 3, 416, 416) and x.dtype == np.float32 and 0.0 <= x.min() and x.max() <= 1.0
    np.testing.assert_array_equal(x[0, 1], img[:, :, 1].astype(np.float32) / np.float32(255))     # channel order untouched
    np.testing.assert_array_equal(x, od.cv_img_2_ndarray(img))


@pytest.mark.gpu
def test_config0_test_jpg_end_to_end(cuda):
    """licence_plate/test.jpg -> resize 416 -> cv_img_2_ndarray (HIP) -> CarNet (D53 spec, seeded weights) -> predict:
    the image tensor bit-equal to the oracle's, logits within 1e-3 (fp32 path), the predicted row equal to the
    oracle's on the HIP logits, shapes / dtype / ranges, and two runs bit-identical."""
    import torch
    from oracle import forward as of
    from yolo_amd.net import CarNet
    from yolo_amd.detect import Detector, cv_img_2_ndarray
    img = _test_image()
    x = cv_img_2_ndarray(img, device=cuda)
    assert tuple(x.shape) == (1, 3, 416, 416) and x.dtype == torch.float32
    np.testing.assert_array_equal(x.cpu().numpy(), od.cv_img_2_ndarray(img))
    g = og.build_graph(SPEC)
    P = og.init_params(g, seed=0, bn='random')
    size = (416, 416)
    steps = od.init_steps(SPEC['layers'], SPEC['all_anchors'])
    syxhw = od.init_syxhw(size, steps, SPEC['all_anchors'])
    det = Detector(SPEC, size, steps, device=cuda)
    ref = [r.numpy() for r in of.forward_torch(g, P, x.cpu().numpy())]
    for dtype in ('f32', 'bf16'):
        net = CarNet(SPEC, dtype=dtype, device=cuda).load_params(P)
        outs = net(x)
        assert [tuple(o.shape) for o in outs] == [(1, 2704, 3, 30), (1, 676, 3, 30), (1, 169, 3, 30)]
        if dtype == 'f32':
            for o, r in zip(outs, ref):
                np.testing.assert_allclose(o.cpu().numpy(), r, rtol=0, atol=1e-3)
        pred = det.predict(outs)
        assert pred.shape == (1, 30) and pred.dtype == np.float32 and np.isfinite(pred).all() and 0.0 < pred[0, 0] < 1.0
        rpred, ridx = od.predict([o.cpu().numpy() for o in outs], SPEC['slice_point'], size, syxhw)
        np.testing.assert_allclose(pred, rpred, rtol=1e-5, atol=1e-6)
        pred2 = det.predict(net(x))
        np.testing.assert_array_equal(pred, pred2)                                      # deterministic
