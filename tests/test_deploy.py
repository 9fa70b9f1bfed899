"""Deployment seam: /YOLO/box row (CPU) and export -> init_executor round trip (GPU)."""
import math

import numpy as np
import pytest

from yolo_amd import deploy


def test_car_box_row_azimuth():
    # car/video_node.py:245-251: softmax over the 24 azimuth logits, circular mean of the class directions
    row = np.zeros((1, 30), np.float32)
    row[0, :6] = [0.9, 0.4, 0.6, 0.2, 0.3, 0.123]
    row[0, 6 + 6] = 20.0                              # class 6 = 90 degrees dominates
    out = deploy.car_box_row(row)
    assert abs(out[5] - math.pi / 2) < 1e-4 and out[0] == np.float32(0.9) and row[0, 5] == np.float32(0.123)
    row[0, 6:] = 0.0                                  # uniform distribution: directions cancel
    c = deploy.car_box_row(row)
    assert abs(sum(deploy._COS_OFFSET)) < 1e-9 and np.isfinite(c[5])
    row[0, 6 + 23] = 30.0                             # class 23 = 345 degrees -> -15 degrees
    assert abs(deploy.car_box_row(row)[5] + 15 * math.pi / 180) < 1e-4


def test_symbol_json_round_trip_and_rejections(tmp_path):
    """export-symbol.json (the half of HybridBlock.export the reference's init_executor builds its graph from,
    yolo_gluon.py:206-208): written from the graph, parsed back into the spec's structure for every spec of the repo; the
    heads are listed fine -> coarse (all_output[::-1], car/utils.py:95); older `attr` keys are read; anything that is not
    the CarNet topology is refused."""
    import copy
    import json
    from oracle import graph as og
    from yolo_amd.spec import NetGraph, darknet53_spec
    from yolo_amd import mxparams
    for spec in (darknet53_spec(), og.spec_micro(), og.spec_car_v1()):
        g = NetGraph(spec)
        sym = deploy.symbol_json(g)
        got = deploy.spec_from_symbol(sym)
        assert got['layers'] == list(spec['layers']) and got['channels'] == list(spec['channels'])
        assert got['slice_point'][-1] == spec['slice_point'][-1]
        assert [len(a) for a in got['all_anchors']] == [len(a) for a in spec['all_anchors']]
        n_conv = sum(1 for n in sym['nodes'] if n['op'] == 'Convolution')
        assert n_conv == len(g.convs()) and len(sym['heads']) == len(spec['all_anchors'])
        # every parameter of the .params file is a variable node of the symbol, and nothing else is
        names = {n['name'] for n in sym['nodes'] if n['op'] == 'null'} - {'data'}
        from yolo_amd import mxparams
        assert names == set(mxparams.gluon_param_names(g).values())
        # fine -> coarse: the first head's output conv sees the largest map (the fewest stride-2 convs upstream... the
        # deepest head is built first, so its reshape node comes first in the file and LAST in `heads`)
        assert sym['heads'][-1][0] < sym['heads'][0][0]
    # CarLPNet (car_and_LP/YOLO.py:62-95, exported at :386): the LP branch is one more head -- transposed, not reshaped --
    # and comes back as LP_slice_point
    lp = dict(og.spec_micro(), LP_slice_point=[1, 3, 4, 7, 10])
    g = NetGraph(lp)
    sym = deploy.symbol_json(g)
    assert len(sym['heads']) == 4 and sym['nodes'][sym['heads'][-1][0]]['op'] == 'transpose'
    got = deploy.spec_from_symbol(sym)
    assert got['LP_slice_point'] == [10] and got['layers'] == list(lp['layers']) and got['channels'] == list(lp['channels'])
    assert sum(1 for n in sym['nodes'] if n['op'] == 'Convolution') == len(g.convs()) == len(NetGraph(og.spec_micro()).convs()) + 31
    assert {n['name'] for n in sym['nodes'] if n['op'] == 'null'} - {'data'} == set(mxparams.gluon_param_names(g).values())
    # the branch reads the INPUT of the finest detection block (the concat), not its output
    lp_first = next(n for n in sym['nodes'] if n['op'] == 'Convolution' and 'yolodetectionblockv33_' in n['name'])
    assert sym['nodes'][lp_first['inputs'][0][0]]['op'] == 'Concat'
    spec = og.spec_micro()
    sym = deploy.symbol_json(NetGraph(spec))
    old = copy.deepcopy(sym)
    for n in old['nodes']:
        if 'attrs' in n:
            n['attr'] = n.pop('attrs')
    path = tmp_path / 'export-symbol.json'
    path.write_text(json.dumps(old))
    assert deploy.spec_from_symbol(str(path))['channels'] == list(spec['channels'])
    bad = copy.deepcopy(sym)
    conv = [n for n in bad['nodes'] if n['op'] == 'Convolution'][5]
    conv['attrs']['num_filter'] = str(int(conv['attrs']['num_filter']) + 8)
    with pytest.raises(ValueError):
        deploy.spec_from_symbol(bad)
    bad = copy.deepcopy(sym)
    bad['nodes'].append({'op': 'Pooling', 'name': 'pool0', 'attrs': {}, 'inputs': [[len(bad['nodes']) - 1, 0, 0]]})
    with pytest.raises(ValueError):
        deploy.spec_from_symbol(bad)


@pytest.mark.gpu
def test_export_and_init_executor(cuda, tmp_path):
    import torch
    from oracle import graph as og
    from yolo_amd.net import CarNet
    spec, size = og.spec_micro(), (64, 96)
    P = og.init_params(og.build_graph(spec), seed=7, bn='random')
    net = CarNet(spec, dtype='bf16', device=cuda).load_params(P)
    x = torch.rand((1, 3) + size, device=cuda)
    ref = [o.clone() for o in net(x)]
    path = deploy.export(net, str(tmp_path), epoch=3)
    assert path.endswith('export-0003.params') and (tmp_path / 'export-symbol.json').exists()
    ex = deploy.init_executor(str(tmp_path), spec, size, device=cuda, step=3)
    out = ex.forward(is_train=False, data=x)
    assert len(out) == 3 and all(torch.equal(a, b) for a, b in zip(out, ref))
    # as the reference does it: no spec, the structure comes from export-symbol.json alone (yolo_gluon.py:204-208)
    ex2 = deploy.init_executor(str(tmp_path), None, size, device=cuda, step=3)
    out2 = ex2.forward(is_train=False, data=x)
    assert all(torch.equal(a, b) for a, b in zip(out2, ref))
    other = dict(spec); other['channels'] = [c * 2 for c in spec['channels']]
    with pytest.raises(ValueError):
        deploy.init_executor(str(tmp_path), other, size, device=cuda, step=3)
    with pytest.raises(ValueError):
        ex.forward(is_train=True, data=x)
    # ... and against the ORACLE, not only against the net that wrote the files: the executor built from the two exported
    # files alone evaluates the oracle's graph on the oracle's parameters (bf16: as good as the rounding-aware oracle)
    from oracle import forward as of
    g = og.build_graph(spec)
    xs = x.cpu().numpy()
    f32 = [r.numpy() for r in of.forward_torch(g, P, xs)]
    sim = [r.numpy() for r in of.forward_torch_bf16sim(g, P, xs)]
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    for o, s_, r in zip(out2, sim, f32):
        e_hip, e_sim = rms(o.cpu().numpy() - r) / r.std(), rms(s_ - r) / r.std()
        assert e_hip < 1.5 * e_sim + 1e-3 and e_hip < 0.015, (e_hip, e_sim)
    # the executors inside the north-star tolerance: exact fp32 and the two split paths (dtype is the executor's `fp16` flag's seam)
    for dt in ('f32', 'bf16x3', 'f16x3'):
        ex = deploy.init_executor(str(tmp_path), None, size, device=cuda, step=3, dtype=dt)
        for o, r in zip(ex.forward(is_train=False, data=x), f32):
            np.testing.assert_allclose(o.cpu().numpy(), r, rtol=0, atol=1e-3)


@pytest.mark.gpu
def test_export_and_init_executor_carlpnet(cuda, tmp_path):
    """car_and_LP/YOLO.py:386 exports CarLPNet; the executor built from export-symbol.json + .params ALONE must recover the
    LP branch and return all_output[::-1] + [LP_output] -- against the oracle's CarLPNet forward."""
    import torch
    from oracle import graph as og, forward as of
    from yolo_amd.net import CarLPNet
    spec, size = dict(og.spec_micro(), LP_slice_point=[1, 3, 4, 7, 10]), (64, 96)
    g = og.build_graph(spec)
    P = og.init_params(g, seed=11, bn='random')
    net = CarLPNet(spec, dtype='f32', device=cuda).load_params(P)
    deploy.export(net, str(tmp_path), epoch=0)
    ex = deploy.init_executor(str(tmp_path), None, size, device=cuda, dtype='f32')
    assert type(ex.net).__name__ == 'CarLPNet'
    x = np.random.default_rng(12).random((2, 3) + size, dtype=np.float32)
    out = ex.forward(is_train=False, data=torch.from_numpy(x).to(cuda))
    routs, rlp = of.forward_torch(g, P, x)
    assert len(out) == 4 and tuple(out[3].shape) == tuple(rlp[0].shape) == (2, 8, 12, 10)
    for o, r in zip(out, list(routs) + list(rlp)):
        np.testing.assert_allclose(o.cpu().numpy(), r.numpy(), rtol=0, atol=1e-3)
    with pytest.raises(ValueError):
        deploy.init_executor(str(tmp_path), og.spec_micro(), size, device=cuda, dtype='f32')      # a spec without the branch
