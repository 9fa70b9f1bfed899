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


# ---- RenderCar (car/render_car.py:52-138, 339-408) on a synthetic sprite set -------------------------------------------
def _sprites(root):
    """Opaque rectangles on transparent canvases, named as the reference's blender renders are
    (...azi<1/100 degree>_ele<1/100 degree>.png under <mode>/<cad>/)."""
    from PIL import Image
    import os
    k = 0
    for mode in ('train', 'valid'):
        for cad in ('cadA', 'cadB'):
            d = os.path.join(root, mode, cad)
            os.makedirs(d)
            for azi in (0, 4500, 9000, 27000):
                im = Image.new('RGBA', (200, 120), (0, 0, 0, 0))
                im.paste((40 + 20 * k, 200 - 10 * k, 90, 255), (30, 25, 170, 95))
                im.save(os.path.join(d, 'car%d_azi%d_ele1000.png' % (k, azi)))
                k += 1


def test_render_car_host_geometry_and_labels(tmp_path):
    """The label box must be the bounding box of what was pasted: checked against the alpha mask the renderer returns
    (an independent measurement of the same geometry), for boxes partly outside the image too."""
    _sprites(str(tmp_path))
    rc = render.RenderCar(160, 256, CLASSES, str(tmp_path), augment=False)
    assert len(rc.rawcar_dataset['train']) == 8 and len(rc.rawcar_dataset['valid']) == 8
    np.random.seed(3)
    fg, mask, lab = rc.render_host(16, 'train')
    assert fg.shape == (16, 3, 160, 256) and mask.shape == fg.shape and lab.shape == (16, 1, 30)
    assert fg.dtype == np.float32 and 0.0 <= fg.min() and fg.max() <= 1.0 and 0.0 <= mask.min() and mask.max() <= 1.0
    for i in range(16):
        cls, y, x, h, w, r = lab[i, 0, :6]
        assert 0 <= cls < 24 and int(np.argmax(lab[i, 0, 6:])) == int(cls) and abs(float(lab[i, 0, 6:].sum()) - 1) < 1e-5
        assert abs(r) <= math.radians(30.0) + 1e-6
        assert 0.2 * 0.9 * 120 / 160 * 0.4 < h < 2.0 and 0.1 < w < 2.0
        ys, xs = np.nonzero(mask[i, 0] > 0)
        # the label's box clipped to the image contains the mask's bounding box ...
        t, b, l, rr = (y - h / 2) * 160, (y + h / 2) * 160, (x - w / 2) * 256, (x + w / 2) * 256
        assert ys.min() >= max(t, 0) - 1.5 and ys.max() <= min(b, 160) + 0.5
        assert xs.min() >= max(l, 0) - 1.5 and xs.max() <= min(rr, 256) + 0.5
        if t >= 0 and l >= 0 and b <= 160 and rr <= 256:
            # fully inside: the label box IS the bounding box of the pasted alpha mask, to the pixel (a box cut by the
            # image edge can lose the corner of the rotated sprite that defined its extent on the other axis)
            assert (ys.min(), ys.max() + 1, xs.min(), xs.max() + 1) == (round(t), round(b), round(l), round(rr))
        # at least 70 % of the box is inside the image on each axis (render_car.py:101-108)
        assert min(b, 160) - max(t, 0) >= 0.69 * (b - t) - 1 and min(rr, 256) - max(l, 0) >= 0.69 * (rr - l) - 1
    # a seeded run is reproducible, and render_rate = 0 renders nothing
    np.random.seed(3)
    fg2, mask2, lab2 = rc.render_host(16, 'train')
    assert np.array_equal(fg, fg2) and np.array_equal(mask, mask2) and np.array_equal(lab, lab2)
    fg0, mask0, lab0 = rc.render_host(4, 'valid', render_rate=0.0)
    assert (lab0 == -1).all() and not fg0.any() and not mask0.any()
    with pytest.raises(NotImplementedError):
        rc.render_host(1, 'train', pascal_rate=0.5)


def test_color_augmenter_known_answers():
    """mxnet's colour augmenters restated (ColorJitterAug -> HueJitterAug -> LightingAug): with every strength at zero the
    chain is the identity; brightness alone scales; hue by a whole turn (alpha = +-1 -> 180 degrees twice = 360) ... the
    YIQ round trip ityiq @ tyiq is the identity to 3 decimals (the published matrices are rounded)."""
    import random
    img = (np.random.default_rng(0).random((5, 7, 3)) * 255).astype(np.float32)
    ident = render.ColorAugmenter(0, 0, 0, 0, 0)
    random.seed(1); np.random.seed(1)
    np.testing.assert_allclose(ident(img), img, rtol=0, atol=1.0)             # (only the rounded YIQ matrices differ from I)
    np.testing.assert_allclose(render.ColorAugmenter.ITYIQ @ render.ColorAugmenter.TYIQ, np.eye(3), atol=2e-3)
    random.seed(5); np.random.seed(5)
    b = render.ColorAugmenter(0.3, 0, 0, 0, 0)
    out = b(img)
    ratio = out / np.maximum(ident(img), 1e-3)
    assert 0.7 - 1e-3 <= float(np.median(ratio)) <= 1.3 + 1e-3 and float(ratio.std()) < 0.02
    random.seed(7); np.random.seed(7)
    full = render.ColorAugmenter()
    o1 = full(img)
    random.seed(7); np.random.seed(7)
    assert np.array_equal(o1, full(img)) and o1.shape == img.shape and o1.dtype == np.float32


@pytest.mark.gpu
def test_render_car_on_device(cuda, tmp_path):
    """RenderCar.render: the host batch composited by yolo_composite equals the reference's blend formula, labels ride
    along; the rendered batch drives a training step."""
    import torch
    _sprites(str(tmp_path))
    rc = render.RenderCar(64, 96, CLASSES, str(tmp_path), device=cuda)
    bg = (torch.rand((4, 3, 64, 96)) * 255).to(cuda)
    np.random.seed(11); import random; random.seed(11)
    img, lab = rc.render(bg, 'train', render_rate=0.75)
    np.random.seed(11); random.seed(11)
    fg, mask, lab_h = rc.render_host(4, 'train', render_rate=0.75)
    ref = np.clip(bg.cpu().numpy() / np.float32(255.) * (1 - mask) + fg * mask, 0, 1)
    np.testing.assert_allclose(img.cpu().numpy(), ref, rtol=0, atol=1e-6)
    np.testing.assert_array_equal(lab.cpu().numpy(), lab_h)
    assert tuple(img.shape) == (4, 3, 64, 96) and tuple(lab.shape) == (4, 1, 30)
