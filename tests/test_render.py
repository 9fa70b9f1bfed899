"""Synthetic-target helpers."""
import math

import numpy as np
import pytest

from yolo_amd import render

CLASSES = [[15.0 * i, 0.0] for i in range(24)]


def test_label_dist_and_label_row():
    """get_label_dist (car/render_car.py:410-438) against answers worked out by hand.  On the equator (elevation 0, all
    24 classes at elevation 0, 15 degrees apart) the great-circle angle to class k is the azimuth difference, so for
    azimuth 0 the distribution is exp(-(min(k, 24 - k) * pi / 12)^2 / 0.1), normalised."""
    c, d = render.get_label_dist(0.0, 0.0, CLASSES)
    k = np.arange(24)
    g = np.exp(-(np.minimum(k, 24 - k) * math.pi / 12) ** 2 / 0.1)
    assert c == 0 and d.dtype == np.float32 and d.shape == (24,)
    np.testing.assert_allclose(d, g / g.sum(), rtol=2e-5, atol=1e-9)
    # the three largest entries as plain numbers: 1, e^-0.68539, e^-2.74156 over their sum 2.1410
    np.testing.assert_allclose(d[[0, 1, 23, 2]], np.array([1.0, 0.503894, 0.503894, 0.064470]) / 2.140962, rtol=1e-4)
    # azimuth 90 degrees = class 6: the same distribution rotated by six classes
    c6, d6 = render.get_label_dist(0.0, math.pi / 2, CLASSES)
    assert c6 == 6
    np.testing.assert_allclose(d6, np.roll(d, 6), rtol=1e-4, atol=1e-9)
    # at the pole every class direction is 90 degrees away: uniform, and the arg-min is the first class
    cp, dp = render.get_label_dist(math.pi / 2, 1.234, CLASSES)
    np.testing.assert_allclose(dp, np.full(24, 1 / 24.0), rtol=1e-5)
    # half way between two classes: they share the top probability
    ch, dh = render.get_label_dist(0.0, math.radians(7.5), CLASSES)
    assert ch in (0, 1) and abs(float(dh[0]) - float(dh[1])) < 1e-6 and float(dh[0]) > float(dh[2])
    for ele, azi in ((0.0, 0.3), (0.1, 3.0), (0.0, 6.2)):
        c, d = render.get_label_dist(ele, azi, CLASSES)
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
