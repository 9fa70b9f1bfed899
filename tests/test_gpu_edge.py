"""Edge shapes of the whole path through the Python mirror (smallest maps, one class, one anchor per scale, wide
class lists, odd batches): forward logits, top-1 index, per-class NMS and one training step's losses against the oracle."""
import numpy as np
import pytest
import torch

from oracle import graph as og, forward as of, detect as od, train as ot

pytestmark = pytest.mark.gpu

MICRO = og.spec_micro()
CASES = {
    'deepest map 1x1, batch 1': (MICRO, (32, 32), 1),
    'one class': (dict(MICRO, slice_point=[1, 3, 5, 6, 7]), (64, 64), 2),
    'one anchor per scale': (dict(MICRO, all_anchors=[[a[0]] for a in MICRO['all_anchors']]), (64, 64), 2),
    '80 classes': (dict(MICRO, slice_point=[1, 3, 5, 6, 86]), (64, 96), 2),
    'odd batch, non-square': (MICRO, (96, 160), 5),
}


@pytest.mark.parametrize('dtype,tol', [('f32', 1e-3), ('bf16', 5e-2)])
@pytest.mark.parametrize('name', sorted(CASES))
def test_edge_shape_end_to_end(cuda, name, dtype, tol):
    from yolo_amd.net import CarNet
    from yolo_amd.detect import Detector
    from yolo_amd.train import Trainer
    spec, size, B = CASES[name]
    g = og.build_graph(spec)
    P = og.init_params(g, seed=1, bn='random')
    x = np.random.default_rng(3).random((B, 3) + size, dtype=np.float32)
    ref = of.forward_torch(g, P, x)
    net = CarNet(spec, dtype=dtype, device=cuda).load_params(P)
    outs = net(torch.from_numpy(x).to(cuda))
    for o, r in zip(outs, ref):
        assert float(np.abs(o.cpu().numpy() - r.numpy()).max()) < tol
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])
    det = Detector(spec, size, steps, device=cuda)
    host = [o.cpu().numpy() for o in outs]
    pred, idx = det.predict_device(outs)
    rpred, ridx = od.predict(host, spec['slice_point'], size, syxhw)          # the oracle on THESE logits: exact indices
    assert idx.cpu().tolist() == ridx.tolist()
    np.testing.assert_allclose(pred.cpu().numpy(), rpred, rtol=1e-5, atol=1e-6)
    rows, scores = det.decode_scores(outs, 'class')
    kept, _, cnt = det.nms(rows, 'class', scores=scores)
    rk, _ = od.nms(rows[0].cpu().numpy(), 'class', scores=scores[0].cpu().numpy())
    assert kept[0, :int(cnt[0])].cpu().tolist() == rk.tolist()
    if dtype == 'f32':
        lab = ot.synthetic_labels(B, seed=5, render_rate=0.0, num_class=spec['slice_point'][-1] - 6)
        tr = Trainer(CarNet(spec, dtype=dtype, device=cuda).load_params(P), size)
        losses = tr.train_step(torch.from_numpy(x).to(cuda), torch.from_numpy(lab).to(cuda))
        rl, _, _ = ot.train_step_reference(g, P, x, lab, spec, size)
        np.testing.assert_allclose(losses.cpu().numpy(), np.stack(rl), rtol=1e-3, atol=1e-6)


def _row(ncls, c, y, x, h, w, r=0.1):
    v = -np.ones(6 + ncls, np.float32)
    v[:6] = [c, y, x, h, w, r]
    v[6:] = 0.0
    v[6 + int(c)] = 1.0
    return v


def _label_cases():
    n = MICRO['slice_point'][-1] - 6
    none = -np.ones(6 + n, np.float32)
    S = (64, 96)
    return {
        'no object in any image': (S, [[none], [none]], None),
        'object only in one image': (S, [[none], [_row(n, 1, .5, .5, .4, .3)]], None),
        'two objects in one cell': (S, [[_row(n, 0, .5, .5, .4, .3), _row(n, 2, .51, .51, .38, .31)], [_row(n, 1, .3, .7, .2, .2), none]], None),
        'objects on the border and the corner': (S, [[_row(n, 0, .0, .0, .3, .3)], [_row(n, 3, 1.0, 1.0, .5, .5)]], None),
        'tiny and oversized boxes': (S, [[_row(n, 0, .5, .5, .01, .01)], [_row(n, 1, .5, .5, 1.5, 1.5)]], None),
        'batch 1 at 32x32 (BatchNorm over one pixel)': ((32, 32), [[_row(n, 1, .5, .5, .4, .3)]], None),
        'constant image (zero variance in the stem)': (S, [[_row(n, 1, .5, .5, .4, .3)], [_row(n, 2, .4, .6, .3, .3)]], 0.5),
        'rotation at +-pi': (S, [[_row(n, 0, .5, .5, .4, .3, r=np.pi)], [_row(n, 1, .5, .5, .4, .3, r=-np.pi)]], None),
    }


@pytest.mark.parametrize('name', sorted(_label_cases()))
def test_training_step_label_and_input_edges(cuda, name):
    """One fp32 training step (update off) on label / input edge cases: the five losses and EVERY parameter gradient against the
    oracle's autograd restatement."""
    from yolo_amd.net import CarNet
    from yolo_amd.train import Trainer
    size, labels, xconst = _label_cases()[name]
    labels = np.asarray(labels, np.float32)
    B = labels.shape[0]
    g = og.build_graph(MICRO)
    P = og.init_params(g, seed=1, bn='random')
    x = np.random.default_rng(3).random((B, 3) + size, dtype=np.float32)
    if xconst is not None:
        x[:] = xconst
    tr = Trainer(CarNet(MICRO, dtype='f32', device=cuda).load_params(P), size)
    losses = tr.train_step(torch.from_numpy(x).to(cuda), torch.from_numpy(labels).to(cuda), update=False)
    rl, rg, _ = ot.train_step_reference(g, P, x, labels, MICRO, size)
    np.testing.assert_allclose(losses.cpu().numpy(), np.stack(rl), rtol=1e-4, atol=1e-6)
    for n, gr in tr.grads().items():
        ref = rg[n]
        den = max(float(np.abs(ref).max()), 1e-6)
        assert float(np.abs(gr.cpu().numpy() - ref).max()) / den < 1e-2, n
