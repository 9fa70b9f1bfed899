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
    path = deploy.export_params(net, str(tmp_path), epoch=3)
    assert path.endswith('export-0003.params')
    ex = deploy.init_executor(str(tmp_path), spec, size, device=cuda, step=3)
    out = ex.forward(is_train=False, data=x)
    assert len(out) == 3 and all(torch.equal(a, b) for a, b in zip(out, ref))
    with pytest.raises(ValueError):
        ex.forward(is_train=True, data=x)
