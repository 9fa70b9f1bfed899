"""GPU path against the committed golden fixtures (tests/golden/*.npz): fixed inputs and expected
outputs that do not depend on the oracle code running at test time."""
import os

import numpy as np
import pytest
import torch

from oracle import graph as og, detect as od      # only for the seeded parameter / input streams and specs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('name,spec,size', [('forward_micro_identity', og.spec_micro(), (64, 96)),
                                            ('forward_micro_random', og.spec_micro(), (64, 96)),
                                            ('forward_test_yaml', og.spec_test_yaml(), (192, 256))])
@pytest.mark.parametrize('tune', ['auto', 'measure'])
def test_forward_f32_vs_golden(cuda, name, spec, size, tune):
    from yolo_amd.net import CarNet
    z = np.load(os.path.join(GOLD, name + '.npz'))
    B, _, _, seed_p, seed_x = [int(v) for v in z['meta']]
    P = og.init_params(og.build_graph(spec), seed=seed_p, bn=str(z['bn']))
    x = np.random.default_rng(seed_x).random((B, 3) + size, dtype=np.float32)
    net = CarNet(spec, dtype='f32', device=cuda, tune=tune).load_params(P)
    outs = net(torch.from_numpy(x).to(cuda))
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.cpu().numpy(), z['out%d' % i], rtol=0, atol=1e-3)   # north_star: 1e-3 fp32


def test_detect_vs_golden(cuda):
    from yolo_amd.detect import Detector, get_iou
    z = np.load(os.path.join(GOLD, 'detect_416.npz'))
    spec, size = og.spec_d53(), (416, 416)
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    rng = np.random.default_rng(11)
    outs = [torch.from_numpy((1.5 * rng.standard_normal((2, a, 3, 30))).astype(np.float32)).to(cuda) for a in area]
    det = Detector(spec, size, steps, device=cuda)
    rows = det.decode(outs)
    np.testing.assert_allclose(rows.cpu().numpy()[:, z['sel']], z['rows_sel'], rtol=1e-5, atol=1e-6)
    pred, idx = det.predict_device(outs)
    assert idx.cpu().tolist() == z['idx'].tolist()                      # bit-exact index
    np.testing.assert_allclose(pred.cpu().numpy(), z['pred'], rtol=1e-5, atol=1e-6)
    k, s, c = det.nms(rows, 'obj')
    assert k[0, :int(c[0])].cpu().tolist() == z['kept_obj'].tolist()    # bit-exact kept ids
    k, s, c = det.nms(rows, 'class')
    assert k[0, :int(c[0])].cpu().tolist() == z['kept_cls'].tolist()
    np.testing.assert_allclose(s[0, :int(c[0])].cpu().numpy(), z['score_cls'], rtol=2e-5)
    ltrb = od.get_default_ltrb(size, steps, spec['all_anchors'])
    iou = get_iou(torch.from_numpy(ltrb).to(cuda), torch.from_numpy(z['target']), mode=2).cpu().numpy().reshape(-1)
    assert int(np.argmax(iou)) == int(z['iou_argmax'][0])
    np.testing.assert_allclose(iou[z['sel']], z['iou_sel'], rtol=1e-6, atol=1e-7)


def test_lp_branch_vs_golden(cuda):
    """CarLPNet fixture (tests/golden/lp_micro.npz): LP-branch output, predict_LP rows, LP cells and losses."""
    import ctypes as C
    from yolo_amd.net import CarLPNet
    from yolo_amd.detect import predict_LP_batch
    from yolo_amd import lib as L
    z = np.load(os.path.join(GOLD, 'lp_micro.npz'))
    spec = dict(og.spec_micro(), LP_slice_point=[1, 3, 4, 7, 10], LP_r_max=[45, 60, 45])
    size = (64, 96)
    P = og.init_params(og.build_graph(spec), seed=3, bn='random')
    x = np.random.default_rng(4).random((3, 3) + size, dtype=np.float32)
    net = CarLPNet(spec, dtype='f32', device=cuda).load_params(P)
    outs, lp = net(torch.from_numpy(x).to(cuda))
    np.testing.assert_allclose(lp[0].cpu().numpy(), z['lp_out'], rtol=0, atol=1e-3)
    # post-processing and losses on the FIXTURE's logits: exact cells, tight values
    gold = torch.from_numpy(z['lp_out']).to(cuda)
    pred = predict_LP_batch([gold], spec['LP_slice_point'], spec['LP_r_max'])
    np.testing.assert_allclose(pred, z['pred'], rtol=1e-6, atol=1e-6)
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    labels = torch.from_numpy(z['lp_labels']).to(cuda)
    rec = torch.empty((3, 1, 8 + 3), device=cuda)
    step = od.init_steps(spec['layers'], spec['all_anchors'])[0]
    assert lib.yolo_assign_targets_lp(labels.data_ptr(), rec.data_ptr(), 3, 1, 10, 3, size[0], size[1], step, 45.0, 60.0, 45.0, st) == 0
    fw = size[1] // step
    assert [int(v) for v in rec[:, 0, 1].cpu()] == [int(h * fw + w) for h, w in z['cells']]      # bit-exact cells
    logits = gold.reshape(3, -1, 10).contiguous()
    dl = torch.empty_like(logits); ls = torch.empty((5, 3), device=cuda)
    s5 = (C.c_float * 5)(0.1, 10.0, 1.0, 0.1, 0.3)
    assert lib.yolo_loss_lp_fwd_bwd(logits.data_ptr(), rec.data_ptr(), dl.data_ptr(), ls.data_ptr(), 3, logits.shape[1], 10, 1, s5, 1.0, 0.1, st) == 0
    np.testing.assert_allclose(ls.cpu().numpy(), z['losses'], rtol=1e-4, atol=1e-8)
    g = dl.cpu().numpy()
    np.testing.assert_allclose(np.abs(g).sum(), z['grad_sum'][0], rtol=1e-4)
    np.testing.assert_allclose(g.reshape(-1)[::37], z['grad_sel'], rtol=1e-3, atol=1e-7)


def test_predict_lp_vs_the_reference_itself(cuda):
    """SURVEY row a21 against outputs of the reference's OWN `predict_LP` (numpy branch, licence_plate/LP_detection.py:147-162), run in
    the build container by tests/golden/make_reference_vectors.py: the HIP kernel picks the same cell (first index among ties) and
    returns the same pose row (device expf against numpy's: a few float32 ulps)."""
    from yolo_amd.detect import predict_LP
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors.npz'))
    for k in range(int(G['lp_cases'])):
        x, want, r_max = G['lp_in_%d' % k], G['lp_out_%d' % k], [float(v) for v in G['lp_rmax_%d' % k]]
        got = predict_LP(torch.from_numpy(x).to(cuda), r_max)
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-7, err_msg='case %d' % k)
