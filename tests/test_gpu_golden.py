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
    iou = get_iou(torch.from_numpy(ltrb).to(cuda), torch.from_numpy(z['target'])).cpu().numpy().reshape(-1)
    assert int(np.argmax(iou)) == int(z['iou_argmax'][0])
    np.testing.assert_allclose(iou[z['sel']], z['iou_sel'], rtol=1e-6, atol=1e-7)
