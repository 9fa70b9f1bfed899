"""Synthetic-target helpers."""
import math

import numpy as np
import pytest

from yolo_amd import render

CLASSES = [[15.0 * i, 0.0] for i in range(24)]


def test_label_dist_and_label_row():
    from oracle import train as ot
    for ele, azi in ((0.0, 0.3), (0.1, 3.0), (0.0, 6.2)):
        c, d = render.get_label_dist(ele, azi, CLASSES)
        c2, d2 = ot.get_label_dist(ele, azi, CLASSES)
        assert c == c2 and np.array_equal(d, d2)
        assert abs(float(d.sum()) - 1.0) < 1e-6 and int(np.argmax(d)) == c
    c, d = render.get_label_dist(0.0, math.radians(44.0), CLASSES)
    assert c == 3                                                       # nearest of 0, 15, 30, 45, ...
    (xlo, xhi), (ylo, yhi) = render.paste_range(10, 20, 110, 80, 320, 512)
    assert (xlo, xhi, ylo, yhi) == (-40, 432, -38, 258)
    lab = render.car_label(3, 10, 20, 110, 80, paste_x=100, paste_y=50, r=0.2, label_distribution=d, img_h=320, img_w=512)
    assert lab.shape == (1, 30)
    np.testing.assert_allclose(lab[0, :6], [3, (50 + 50) / 320., (60 + 100) / 512., 60 / 320., 100 / 512., 0.2], rtol=1e-6)
    assert np.array_equal(lab[0, 6:], d)
    assert (render.empty_labels(4, 24) == -1).all()


@pytest.mark.gpu
def test_composite(cuda):
    import torch
    g = torch.Generator(device='cpu').manual_seed(0)
    bg = (torch.rand((2, 3, 20, 28), generator=g) * 300 - 20).to(cuda)      # out-of-range values exercise the clip
    fg = torch.rand((2, 3, 20, 28), generator=g).to(cuda)
    mask = (torch.rand((2, 3, 20, 28), generator=g) > 0.5).float().to(cuda)
    out = render.composite(bg, fg, mask)
    ref = torch.clamp((bg / 255.) * (1 - mask) + fg * mask, 0, 1)
    assert torch.allclose(out, ref, rtol=0, atol=1e-6)
