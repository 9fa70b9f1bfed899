"""CPU tests of the oracle: the two restatements agree, analytic known answers hold, the structure
matches what the reference author recorded in comments, and the committed golden fixtures reproduce."""
import os

import numpy as np
import pytest
import torch

from oracle import graph as og, forward as of, detect as od, train as ot

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_structure_matches_reference_comments():
    # car/YOLO.py:661-662: car/v1 spec at 320x512 -> (1,640,3,30),(1,160,3,30),(1,40,3,30)
    steps = od.init_steps(og.spec_car_v1()['layers'], og.spec_car_v1()['all_anchors'])
    assert steps == [16, 32, 64]
    assert od.init_area((320, 512), steps) == [640, 160, 40]
    # car/YOLO.py:135 comment "[12*16,6*8,3*4]" for test.yaml at 192x256
    steps = od.init_steps(og.spec_test_yaml()['layers'], og.spec_test_yaml()['all_anchors'])
    assert od.init_area((192, 256), steps) == [12 * 16, 6 * 8, 3 * 4]
    # parameter and FLOP counts of the reference-structured graphs
    g = og.build_graph(og.spec_d53())
    assert len(og.conv_list(g)) == 75
    assert abs(og.count_params(g) / 1e6 - 123.12) < 0.01
    assert abs(og.conv_flops(g, 416, 416) / 1e9 - 113.26) < 0.01
    assert abs(og.conv_flops(g, 608, 608) / 1e9 - 241.94) < 0.01
    gc = og.build_graph(og.spec_car_v1())
    assert len(og.conv_list(gc)) == 88 and abs(og.conv_flops(gc, 320, 512) / 1e9 - 29.77) < 0.01


def test_two_restatements_agree():
    spec = og.spec_micro()
    g = og.build_graph(spec)
    P = og.init_params(g, seed=3, bn='random')
    x = np.random.default_rng(4).random((2, 3, 64, 96), dtype=np.float32)
    a = of.forward_torch(g, P, x)
    b = of.forward_numpy64(g, P, x)
    assert [tuple(t.shape) for t in a] == [(2, 96, 3, 10), (2, 24, 3, 10), (2, 6, 3, 10)]
    for u, v in zip(a, b):
        assert np.abs(u.numpy() - v).max() < 1e-5


@pytest.mark.parametrize('name,spec,size', [('forward_micro_identity', og.spec_micro(), (64, 96)),
                                            ('forward_micro_random', og.spec_micro(), (64, 96)),
                                            ('forward_test_yaml', og.spec_test_yaml(), (192, 256))])
def test_forward_golden(name, spec, size):
    z = np.load(os.path.join(GOLD, name + '.npz'))
    B, _, _, seed_p, seed_x = [int(v) for v in z['meta']]
    g = og.build_graph(spec)
    P = og.init_params(g, seed=seed_p, bn=str(z['bn']))
    x = np.random.default_rng(seed_x).random((B, 3) + size, dtype=np.float32)
    outs = of.forward_torch(g, P, x)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), z['out%d' % i], rtol=0, atol=2e-5)


def test_zero_logits_known_answer():
    spec, size = og.spec_d53(), (416, 416)
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    outs = [np.zeros((1, a, 3, 30), np.float32) for a in area]
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])
    rows = od.decode_all(outs, spec['slice_point'], size, syxhw)[0]
    assert np.all(rows[:, 0] == 0.5)
    ltrb = od.get_default_ltrb(size, steps, spec['all_anchors']).reshape(-1, 4)
    np.testing.assert_allclose(rows[:, 1:5], ltrb, atol=1e-6)       # decode(0) == the anchor boxes of _get_default_ltrb
    pred, idx = od.predict(outs, spec['slice_point'], size, syxhw)
    assert idx[0] == 0                                                # all tied -> lowest index


def test_iou_known_answers():
    box = np.asarray([[0.1, 0.2, 0.4, 0.6]], np.float32)
    assert abs(float(od.get_iou(box, np.asarray([0, 0.4, 0.25, 0.4, 0.3], np.float32))[0, 0]) - 1) < 1e-6
    assert float(od.get_iou(box, np.asarray([0, 0.9, 0.9, 0.05, 0.05], np.float32))[0, 0]) == 0
    assert od.box_iou_ltrb(box[0], box[0]) == 1.0


def test_detect_golden():
    z = np.load(os.path.join(GOLD, 'detect_416.npz'))
    spec, size = og.spec_d53(), (416, 416)
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    rng = np.random.default_rng(11)
    outs = [(1.5 * rng.standard_normal((2, a, 3, 30))).astype(np.float32) for a in area]
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])
    rows = od.decode_all(outs, spec['slice_point'], size, syxhw)
    np.testing.assert_allclose(rows[:, z['sel']], z['rows_sel'], rtol=1e-6)
    pred, idx = od.predict(outs, spec['slice_point'], size, syxhw)
    assert np.array_equal(idx, z['idx'])
    np.testing.assert_allclose(pred, z['pred'], rtol=1e-6)
    for mode in ('obj', 'cls'):
        k, s = od.nms(rows[0], 'obj' if mode == 'obj' else 'class')
        assert np.array_equal(k, z['kept_' + mode])
        np.testing.assert_allclose(s, z['score_' + mode], rtol=1e-6)
    assert z['kept_obj'][0] == z['idx'][0]              # kept[0] == argmax(sigmoid(obj))


def test_nms_semantics():
    # three boxes: 0 and 1 overlap heavily (IoU > 0.45), 2 is disjoint; scores 0.9, 0.8, 0.7
    rows = np.zeros((3, 8), np.float32)
    rows[:, 0] = [0.9, 0.8, 0.7]
    rows[:, 1:5] = [[0.1, 0.1, 0.5, 0.5], [0.12, 0.1, 0.52, 0.5], [0.6, 0.6, 0.9, 0.9]]
    k, s = od.nms(rows, 'obj')
    assert k.tolist() == [0, 2]
    rows[:, 6:] = [[9, -9], [-9, 9], [9, -9]]            # class mode: box 1 is another class -> survives
    k, _ = od.nms(rows, 'class')
    assert sorted((k // 2).tolist()) == [0, 1, 2]
    rows[:, 0] = 0.5                                     # ties -> lower candidate id first
    k, _ = od.nms(rows[:, :6].copy(), 'obj', iou_thresh=0.99) if False else od.nms(rows, 'obj', iou_thresh=0.99)
    assert k.tolist() == [0, 1, 2]


def test_train_golden():
    z = np.load(os.path.join(GOLD, 'train_micro.npz'))
    spec, size = og.spec_micro(), (64, 96)
    g = og.build_graph(spec)
    P = og.init_params(g, seed=0, bn='random')
    x = np.random.default_rng(2).random((2, 3) + size, dtype=np.float32)
    torch.manual_seed(0)
    losses, grads, merged = ot.train_step_reference(g, P, x, z['labels'], spec, size)
    np.testing.assert_allclose(np.stack(losses), z['losses'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(grads['stem.weight'], z['grad_stem_w'], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(grads['heads.0.out.bias'], z['grad_out_bias'], rtol=1e-4, atol=1e-7)
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    ltrb = od.get_default_ltrb(size, steps, spec['all_anchors'])
    px, anc, box = ot.find_best(z['labels'][0, 0], ltrb, spec['all_anchors'], size, steps, area)
    assert [px, anc] == z['find_best'].tolist()
    np.testing.assert_allclose(box, z['box'], rtol=1e-6)


def test_assignment_inverse_of_decode():
    """C.3: the target encode inverts the decode.  (IoU ties are the rule, not the exception -- whenever one
    box contains the other along an axis every contained position has the same overlap -- and the
    reference then takes the FIRST arg-max and clips sigmoid(t) to [1e-4, .9999]; the exact inverse holds
    when the arg-max is unique, e.g. a label that coincides with an anchor box shifted inside its cell.)"""
    spec, size = og.spec_d53(), (416, 416)
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    ltrb = od.get_default_ltrb(size, steps, spec['all_anchors'])
    px0, anc0 = area[0] + 5 * 26 + 7, 1                       # scale 1 (stride 16), cell (5,7), anchor 1
    l, t, r, b = ltrb[px0, anc0]
    L = np.asarray([2, (t + b) / 2, (l + r) / 2, b - t, r - l, 0.1] + [0] * 24, np.float32)
    px, anc, box = ot.find_best(L, ltrb, spec['all_anchors'], size, steps, area)
    assert (px, anc) == (px0, anc0)
    np.testing.assert_allclose(box, 0, atol=2e-5)             # sigmoid^-1(.5) = 0, log(1) = 0
    L2 = L.copy(); L2[1] += 3.0 / 416; L2[2] -= 5.0 / 416; L2[3] *= 1.1; L2[4] *= 0.9     # still the unique best match
    px, anc, box = ot.find_best(L2, ltrb, spec['all_anchors'], size, steps, area)
    assert (px, anc) == (px0, anc0)
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])
    yxhw = np.zeros((1, sum(area), 3, 4), np.float32)
    yxhw[0, px, anc] = box
    out = od.yxhw_to_ltrb(yxhw, size, syxhw)[0, px, anc]
    np.testing.assert_allclose([(out[1] + out[3]) / 2, (out[0] + out[2]) / 2, out[3] - out[1], out[2] - out[0]],
                               L2[1:5], rtol=1e-4)
    # and the clipped case the reference produces for a centre outside the assigned cell
    L3 = np.asarray([2, 0.43, 0.57, 0.31, 0.22, 0.1] + [0] * 24, np.float32)
    _, _, box3 = ot.find_best(L3, ltrb, spec['all_anchors'], size, steps, area)
    assert abs(box3[0] - ot.inv_sigmoid(np.float32(0.9999))) < 1e-3


def test_adam_mxnet_formula():
    z = np.load(os.path.join(GOLD, 'train_micro.npz'))
    rng = np.random.default_rng(7)
    w = rng.standard_normal(16).astype(np.float32); m = np.zeros(16, np.float32); v = np.zeros(16, np.float32)
    for t in range(1, 4):
        ot.adam_step(w, rng.standard_normal(16).astype(np.float32), m, v, t, lr=1e-3, rescale=1.0 / 64)
        np.testing.assert_allclose(w, z['adam'][t - 1], rtol=1e-6)
    # first step moves every weight by ~lr regardless of gradient scale (epsilon outside the correction)
    w0 = np.zeros(4, np.float32)
    ot.adam_step(w0, np.asarray([1e-3, -2.0, 5.0, -1e-2], np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32), 1)
    np.testing.assert_allclose(np.abs(w0), 1e-3, rtol=1e-3)


def test_plumbing_config1():
    """BASELINE config 1: image -> (1,3,H,W)/255 tensor plumbing + LPD predict_LP decode."""
    img = np.random.default_rng(5).integers(0, 256, (246, 560, 3), dtype=np.uint8)
    t = od.cv_img_2_ndarray(img)
    assert t.shape == (1, 3, 246, 560) and t.dtype == np.float32 and 0 <= t.min() and t.max() <= 1
    out = np.random.default_rng(6).standard_normal((1, 10, 10, 16)).astype(np.float32)
    pred, best = od.predict_LP(out, r_max=[40, 40, 40])
    assert best == int(np.argmax(out[0, 0].reshape(-1)))
    assert 0 < pred[0] < 1 and np.all(np.abs(pred[4:7]) <= 40 * np.pi / 180 + 1e-6)
    assert ot.split_render_data(list(range(10)), 4) == [[0, 1], [2, 3, 4], [5, 6], [7, 8, 9]]


def test_lp_branch_two_restatements_agree_and_predict_lp():
    """CarLPNet's LP branch (car_and_LP/YOLO.py:47-95): the torch-fp32 and numpy-fp64 restatements agree, and
    predict_LP picks the cell with the highest sigmoid(score) and applies LP_pose_activation."""
    import math
    from oracle import graph as og, forward as of, detect as od
    spec = dict(og.spec_micro(), LP_slice_point=[1, 3, 4, 7, 10])
    g = og.build_graph(spec)
    assert len(og.conv_list(g)) == len(og.conv_list(og.build_graph(og.spec_micro()))) + 31
    P = og.init_params(g, 1, 'random')
    x = np.random.default_rng(2).random((2, 3, 64, 64), dtype=np.float32)
    (o1, l1), (o2, l2) = of.forward_torch(g, P, x), of.forward_numpy64(g, P, x)
    assert l1[0].shape == (2, 8, 8, 10)
    assert max(np.abs(a.numpy() - b).max() for a, b in zip(o1, o2)) < 1e-5
    assert np.abs(l1[0].numpy() - l2[0]).max() < 1e-5
    lp = np.zeros((1, 2, 3, 10), np.float32)
    lp[0, 1, 2, :7] = [2.0, 0.01, -0.02, 0.003, 0.0, 1.0, -1.0]
    lp[0, 0, 0, 0] = 1.0
    pred, best = od.predict_LP_batch([lp], [1, 3, 4, 7, 10], [45, 60, 45])
    assert best.tolist() == [5] and pred.shape == (1, 7)
    sg = lambda v: 1.0 / (1.0 + math.exp(-v))
    expect = [sg(2.0), 10.0, -20.0, 3.0, 0.0, (sg(1.0) - 0.5) * 2 * 60 * math.pi / 180, (sg(-1.0) - 0.5) * 2 * 45 * math.pi / 180]
    np.testing.assert_allclose(pred[0], expect, rtol=1e-5, atol=1e-6)


def test_lp_golden():
    """The committed LP fixture reproduces from the oracle (make_golden.py: lp_fixture)."""
    from oracle import graph as og, forward as of, detect as od, train as ot
    z = np.load(os.path.join(GOLD, 'lp_micro.npz'))
    spec = dict(og.spec_micro(), LP_slice_point=[1, 3, 4, 7, 10], LP_r_max=[45, 60, 45])
    size = (64, 96)
    g = og.build_graph(spec)
    P = og.init_params(g, seed=3, bn='random')
    x = np.random.default_rng(4).random((3, 3) + size, dtype=np.float32)
    outs, lp = of.forward_torch(g, P, x)
    np.testing.assert_allclose(lp[0].numpy(), z['lp_out'], rtol=0, atol=1e-5)
    pred, best = od.predict_LP_batch([z['lp_out']], spec['LP_slice_point'], spec['LP_r_max'])
    np.testing.assert_array_equal(best, z['best'])
    np.testing.assert_allclose(pred, z['pred'], rtol=1e-6)
    assert np.array_equal(ot.synthetic_lp_labels(3, size, seed=2, add_rate=1.0), z['lp_labels'])
