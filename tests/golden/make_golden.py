#!/usr/bin/env python
"""Regenerates the golden fixtures in this directory from the CPU oracle (run in the build container:
`python tests/golden/make_golden.py`).  The reference itself cannot be run (mxnet / gluoncv absent,
Python-2 sources), so these vectors pin the ORACLE (two independent restatements must agree before a
vector is written) and give the GPU tests fixed, version-independent inputs/outputs.
Weights are never stored: they are re-drawn from numpy default_rng seeds (oracle.graph.init_params)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import graph as og, forward as of, detect as od, train as ot   # noqa: E402


def forward_fixture(name, spec, size, B, seed_p, bn, check64):
    g = og.build_graph(spec)
    P = og.init_params(g, seed=seed_p, bn=bn)
    x = np.random.default_rng(2).random((B, 3) + size, dtype=np.float32)
    outs = [o.numpy() for o in of.forward_torch(g, P, x)]
    if check64:
        o64 = of.forward_numpy64(g, P, x)
        for a, b in zip(outs, o64):
            assert np.abs(a - b).max() < 1e-5, 'restatements disagree'
    np.savez_compressed(os.path.join(HERE, name + '.npz'), out0=outs[0], out1=outs[1], out2=outs[2],
                        meta=np.asarray([B, size[0], size[1], seed_p, 2]), bn=bn)
    return outs


def detect_fixture():
    spec, size = og.spec_d53(), (416, 416)
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    rng = np.random.default_rng(11)
    outs = [(1.5 * rng.standard_normal((2, a, 3, 30))).astype(np.float32) for a in area]
    syxhw = od.init_syxhw(size, steps, spec['all_anchors'])
    rows = od.decode_all(outs, spec['slice_point'], size, syxhw)
    pred, idx = od.predict(outs, spec['slice_point'], size, syxhw)
    k_obj, s_obj = od.nms(rows[0], 'obj')
    k_cls, s_cls = od.nms(rows[0], 'class')
    sel = np.r_[0:64, 5000:5064, 10583:10647]
    ltrb = od.get_default_ltrb(size, steps, spec['all_anchors'])
    target = np.asarray([3, 0.41, 0.52, 0.33, 0.27], np.float32)
    iou = od.get_iou(ltrb, target, mode=2)
    np.savez_compressed(os.path.join(HERE, 'detect_416.npz'), rows_sel=rows[:, sel], sel=sel, pred=pred, idx=idx,
                        kept_obj=k_obj, score_obj=s_obj, kept_cls=k_cls, score_cls=s_cls,
                        iou_argmax=np.asarray([int(np.argmax(iou.reshape(-1)))]), iou_sel=iou.reshape(-1)[sel],
                        target=target)


def train_fixture():
    spec, size = og.spec_micro(), (64, 96)
    g = og.build_graph(spec)
    P = og.init_params(g, seed=0, bn='random')
    x = np.random.default_rng(2).random((2, 3) + size, dtype=np.float32)
    lab = ot.synthetic_labels(2, seed=1, render_rate=0.0, num_class=4)
    losses, grads, merged = ot.train_step_reference(g, P, x, lab, spec, size)
    steps = od.init_steps(spec['layers'], spec['all_anchors'])
    area = od.init_area(size, steps)
    ltrb = od.get_default_ltrb(size, steps, spec['all_anchors'])
    px, anc, box = ot.find_best(lab[0, 0], ltrb, spec['all_anchors'], size, steps, area)
    l2, gout, _ = ot.loss_and_grad_wrt_output(merged, lab, spec, size)
    # Adam trajectory (mxnet formula) on a fixed vector
    rng = np.random.default_rng(7)
    w = rng.standard_normal(16).astype(np.float32); m = np.zeros(16, np.float32); v = np.zeros(16, np.float32)
    traj = []
    for t in range(1, 4):
        gr = rng.standard_normal(16).astype(np.float32)
        ot.adam_step(w, gr, m, v, t, lr=1e-3, rescale=1.0 / 64)
        traj.append(w.copy())
    np.savez_compressed(os.path.join(HERE, 'train_micro.npz'), labels=lab, losses=np.stack(losses),
                        grad_stem_w=grads['stem.weight'], grad_out_bias=grads['heads.0.out.bias'],
                        grad_gamma=grads['stages.2.res.1.c2.gamma'], find_best=np.asarray([px, anc]), box=box,
                        grad_out_sum=np.asarray([np.abs(gout).sum()]), grad_out_sel=gout.reshape(-1)[::97],
                        adam=np.stack(traj))


def lp_fixture():
    """CarLPNet (car_and_LP/YOLO.py): LP-branch output of the micro net, predict_LP rows, LP targets and losses."""
    spec = dict(og.spec_micro(), LP_slice_point=[1, 3, 4, 7, 10], LP_r_max=[45, 60, 45])
    size = (64, 96)
    g = og.build_graph(spec)
    P = og.init_params(g, seed=3, bn='random')
    x = np.random.default_rng(4).random((3, 3) + size, dtype=np.float32)
    outs, lp = of.forward_torch(g, P, x)
    o64, lp64 = of.forward_numpy64(g, P, x)
    assert np.abs(lp[0].numpy() - lp64[0]).max() < 1e-5, 'restatements disagree'
    pred, best = od.predict_LP_batch([lp[0].numpy()], spec['LP_slice_point'], spec['LP_r_max'])
    lpl = ot.synthetic_lp_labels(3, size, seed=2, add_rate=1.0)
    step = od.init_steps(spec['layers'], spec['all_anchors'])[0]
    scale = {'LP_score': 0.1, 'LP_xy': 10.0, 'LP_z': 1.0, 'LP_r': 0.1, 'LP_class': 0.3}
    losses, gout, (y, mask) = ot.lp_loss_and_grad_wrt_output(lp[0].numpy(), lpl, size, step, spec['LP_r_max'],
                                                             spec['LP_slice_point'], scale)
    cells = [ot.find_best_LP(lpl[b, 0], size, step, spec['LP_r_max'])[0] for b in range(3)]
    np.savez_compressed(os.path.join(HERE, 'lp_micro.npz'), lp_out=lp[0].numpy(), pred=pred, best=best, lp_labels=lpl,
                        losses=np.stack(losses), grad_sum=np.asarray([np.abs(gout).sum()]), grad_sel=gout.reshape(-1)[::37],
                        cells=np.asarray(cells))


if __name__ == '__main__':
    lp_fixture()
    forward_fixture('forward_micro_identity', og.spec_micro(), (64, 96), 2, 0, 'identity', True)
    forward_fixture('forward_micro_random', og.spec_micro(), (64, 96), 2, 0, 'random', True)
    forward_fixture('forward_test_yaml', og.spec_test_yaml(), (192, 256), 1, 0, 'random', False)
    detect_fixture()
    train_fixture()
    print('ok', sorted(os.listdir(HERE)))
