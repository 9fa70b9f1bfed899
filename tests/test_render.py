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
            # fully inside: the label box is the box of pixels that are non-zero in ANY band (the reference's PIL), i.e. the
            # alpha mask's bounding box plus the rim of transparent pixels into which the bilinear rotation bled colour: at
            # most two pixels wider on a side, never narrower
            got = (ys.min(), ys.max() + 1, xs.min(), xs.max() + 1)
            lab_box = (round(t), round(b), round(l), round(rr))
            assert 0 <= got[0] - lab_box[0] <= 2 and 0 <= lab_box[1] - got[1] <= 2, (got, lab_box)
            assert 0 <= got[2] - lab_box[2] <= 2 and 0 <= lab_box[3] - got[3] <= 2, (got, lab_box)
        # at least 70 % of the box is inside the image on each axis (render_car.py:101-108)
        assert min(b, 160) - max(t, 0) >= 0.69 * (b - t) - 1 and min(rr, 256) - max(l, 0) >= 0.69 * (rr - l) - 1
    # a seeded run is reproducible, and render_rate = 0 renders nothing
    np.random.seed(3)
    fg2, mask2, lab2 = rc.render_host(16, 'train')
    assert np.array_equal(fg, fg2) and np.array_equal(mask, mask2) and np.array_equal(lab, lab2)
    fg0, mask0, lab0 = rc.render_host(4, 'valid', render_rate=0.0)
    assert (lab0 == -1).all() and not fg0.any() and not mask0.any()
    with pytest.raises(ValueError):
        rc.render_host(1, 'train', pascal_rate=0.5)               # no PASCAL3D+ crops were given


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


# ---- against the ORACLE's restatement (oracle/render.py, written from the reference independently of yolo_amd/render.py) ---
def _pascal_set(root):
    """A PASCAL3D+-shaped data set: opaque 'photographs' with one annotated car box each (one image with two cars, which the
    reference skips), annotations as .mat files whose nesting is what render_car.py:440-458 indexes by position."""
    import os
    import scipy.io as sio
    from PIL import Image
    os.makedirs(os.path.join(root, 'car_imagenet_label'))
    rng = np.random.default_rng(5)
    k = 0
    for mode in ('train', 'valid'):
        d = os.path.join(root, 'car_imagenet_' + mode)
        os.makedirs(d)
        for j in range(4):
            name = 'n0%d_%d' % (k, j)
            w, h = int(rng.integers(180, 260)), int(rng.integers(120, 200))
            px = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
            Image.fromarray(px).save(os.path.join(d, name + '.png'))
            nobj = 2 if (mode == 'train' and j == 3) else 1
            objs = np.zeros((1, nobj), dtype=[('class', 'O'), ('bbox', 'O'), ('anchors', 'O'), ('viewpoint', 'O')])
            for o in range(nobj):
                l, t = int(rng.integers(5, 40)), int(rng.integers(5, 30))
                view = np.zeros((1, 1), dtype=[('azimuth_coarse', 'O'), ('elevation_coarse', 'O'), ('azimuth', 'O'), ('elevation', 'O')])
                view[0, 0] = (np.array([[0.0]]), np.array([[0.0]]), np.array([[float(rng.uniform(0, 360))]]), np.array([[float(rng.uniform(-10, 30))]]))
                objs[0, o] = ('car', np.array([[l, t, w - int(rng.integers(5, 40)), h - int(rng.integers(5, 30))]], np.float64), np.zeros((1, 1)), view)
            rec = np.zeros((1, 1), dtype=[('filename', 'O'), ('objects', 'O')])
            rec[0, 0] = (name + '.png', objs)
            sio.savemat(os.path.join(root, 'car_imagenet_label', name + '.mat'), {'record': rec})
            k += 1


def _oracle_pascal(root, mode, classes):
    import os
    import scipy.io as sio
    from PIL import Image
    from oracle import render as orr, train as ot
    out = []
    d = os.path.join(root, 'car_imagenet_' + mode)
    for img in os.listdir(d):
        ele, azi, box, skip = orr.pascal_azi_ele(sio.loadmat(os.path.join(root, 'car_imagenet_label', img.split('.')[0] + '.mat')))
        if skip:
            continue
        cls, dist = ot.get_label_dist(ele, azi, classes)
        out.append((Image.open(os.path.join(d, img)).convert('RGBA'), box, cls, dist))
    return out


@pytest.mark.parametrize('pascal_rate', [0.0, 0.5, 1.0])
def test_render_car_against_the_oracle(tmp_path, pascal_rate):
    """RenderCar.render_host against oracle.render.render_batch under the same seed: the same sprites, scales, angles,
    offsets; labels to float32 rounding, the composited batch to 1e-6 -- for the PNG branch, the PASCAL3D+ branch
    (render_car.py:262-337, annotated boxes carried through resize and the zero-degree rotation) and a mix."""
    from oracle import render as orr
    _sprites(str(tmp_path / 'png'))
    _pascal_set(str(tmp_path / 'pascal'))
    H, W = 160, 256
    rc = render.RenderCar(H, W, CLASSES, str(tmp_path / 'png'), augment=False, pascal_root=str(tmp_path / 'pascal'))
    assert len(rc.pascal_dataset['train']) == 3 and len(rc.pascal_dataset['valid']) == 4      # (the two-car image is skipped)
    bg = (np.random.default_rng(1).random((12, 3, H, W)) * 255).astype(np.float32)
    np.random.seed(21)
    fg, mask, lab = rc.render_host(12, 'train', pascal_rate=pascal_rate, render_rate=0.8)
    got = np.clip(bg / np.float32(255.) * (1 - mask) + fg * mask, 0, 1)
    np.random.seed(21)
    ref_img, ref_lab = orr.render_batch(bg, rc.rawcar_dataset['train'], _oracle_pascal(str(tmp_path / 'pascal'), 'train', CLASSES),
                                        CLASSES, H, W, pascal_rate=pascal_rate, render_rate=0.8)
    assert (ref_lab[:, 0, 0] >= 0).sum() >= 6 and (ref_lab[:, 0, 0] < 0).any()
    np.testing.assert_allclose(lab, ref_lab, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got, ref_img, rtol=0, atol=2e-6)
    if pascal_rate == 1.0:
        # a PASCAL3D+ label is the ANNOTATED box (not the crop's extent), scaled: 20-90 % of the image on its tighter axis
        for row in lab[lab[:, 0, 0] >= 0, 0]:
            assert row[5] == 0.0 and 0.19 < max(row[3], row[4]) <= 0.9 + 1e-6


def _fonts(root):
    """Glyph images named as licence_plate_render/fonts: 0..33 = digits then letters, 34 = the dot."""
    import os
    from PIL import Image
    os.makedirs(root)
    for k in range(35):
        im = Image.new('RGBA', (30, 60), (0, 0, 0, 0))
        im.paste((10 + 6 * k, 240 - 5 * k, (37 * k) % 255, 255), (4 + k % 5, 6, 24, 50 + k % 7))
        im.save(os.path.join(root, '%d.png' % k))


CAMERA = {'image_width': 640, 'image_height': 480,
          'projection_matrix': {'data': [610.0, 0.0, 322.5, 0.0, 0.0, 608.0, 241.25, 0.0, 0.0, 0.0, 1.0, 0.0]}}


def test_plate_camera_and_homography_known_answers():
    """ProjectRectangle6D (licence_plate_render/__init__.py:273-371) as matrices, K (R3 R2 R1 P + T), against (a) the plain
    pinhole answer for a fronto-parallel plate and (b) the reference's closed form (restated in the oracle) for random poses;
    the homography through four points against a map whose answer is known."""
    from oracle import render as orr
    cam = render.PlateCamera(CAMERA)
    Z = 2500.0
    c = cam.corners([0, 0, Z, 0, 0, 0])
    fx, fy, cx, cy = 610.0, 608.0, 322.5, 241.25
    want = [[cx + fx * 199.5 / Z, cy + fy * 84.0 / Z], [cx - fx * 199.5 / Z, cy + fy * 84.0 / Z],
            [cx - fx * 199.5 / Z, cy - fy * 84.0 / Z], [cx + fx * 199.5 / Z, cy - fy * 84.0 / Z]]
    np.testing.assert_allclose(c, want, rtol=1e-6)
    ocam = dict(fx=fx, fy=fy, cx=cx, cy=cy, w=640, h=480)
    rng = np.random.default_rng(3)
    for _ in range(20):
        pose = [rng.uniform(-700, 700), rng.uniform(-500, 500), rng.uniform(1500, 5000)] + list(rng.uniform(-1, 1, 3) * np.radians([45, 60, 45]))
        np.testing.assert_allclose(cam.corners(pose), orr.project_plate(pose, ocam), rtol=2e-5, atol=2e-3)
    # homography: scale by 2, shift by (3, -1) -- affine, so the projective row must come out as (0, 0, 1)
    src = np.float32([[0, 0], [10, 0], [10, 5], [0, 5]])
    M = render.homography(src, src * 2 + np.float32([3, -1]))
    np.testing.assert_allclose(M, [[2, 0, 3], [0, 2, -1], [0, 0, 1]], atol=1e-9)
    # and a genuinely projective one maps its four points onto their targets
    dst = np.float32([[1, 2], [9, 1], [11, 8], [-1, 6]])
    M = render.homography(src, dst)
    p = (M @ np.concatenate([src, np.ones((4, 1), np.float32)], axis=1).T).T
    np.testing.assert_allclose(p[:, :2] / p[:, 2:], dst, atol=1e-5)
    np.testing.assert_allclose(M, orr.perspective_through(src, dst), atol=1e-9)
    assert cam.centre(0, 0, Z, 240, 320) == (cx * 320 / 640., cy * 240 / 480.)


def test_lp_generator_against_the_oracle(tmp_path):
    """LPGenerator.add's host half against oracle.render.add_plates under the same seed (colour augmenter off): the drawn
    glyphs, the 6-D pose, the projected + blurred + noised plate and the (B,1,10) labels."""
    from PIL import Image
    from oracle import render as orr
    _fonts(str(tmp_path / 'fonts'))
    gen = render.LPGenerator(96, 160, str(tmp_path / 'fonts'), CAMERA, augment=False)
    plate, lp_type, glyphs = gen.draw_LP()
    assert plate.size == (380, 160) and lp_type == 0 and len(glyphs) == 7
    assert all(10 <= g[0] <= 33 for g in glyphs[:3]) and all(0 <= g[0] <= 9 and g[0] != 4 for g in glyphs[3:])
    assert glyphs[0][1:] == [7 / 380., 52 / 380.] and glyphs[3][1] == 175 / 380.
    bg = np.random.default_rng(2).random((6, 3, 96, 160)).astype(np.float32)
    np.random.seed(8)
    fg, mask, lab = gen.add_host(6, 96, 160, [45, 60, 45], add_rate=0.8)
    got = np.clip(bg * (1 - mask) + fg * mask, 0, 1)
    font = [Image.open(str(tmp_path / 'fonts' / ('%d.png' % k))).resize((45, 90), Image.BILINEAR) for k in range(34)]
    dot = Image.open(str(tmp_path / 'fonts' / '34.png')).resize((10, 70), Image.BILINEAR)
    np.random.seed(8)
    ref_img, ref_lab = orr.add_plates(bg, [45, 60, 45], font, dot, dict(fx=610.0, fy=608.0, cx=322.5, cy=241.25, w=640, h=480), add_rate=0.8)
    assert (ref_lab[:, 0, 0] > 0).sum() >= 2 and (ref_lab[:, 0, 0] < 0).any()
    np.testing.assert_allclose(lab, ref_lab, rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(got, ref_img, rtol=0, atol=2e-6)
    for row in lab[lab[:, 0, 0] > 0, 0]:
        assert 1500 <= row[3] <= 5000 and abs(row[1]) <= row[3] * 0.3 + 1e-3 and abs(row[2]) <= row[3] * 7 / 30. + 1e-3
        assert abs(row[4]) <= np.radians(45) + 1e-6 and abs(row[5]) <= np.radians(60) + 1e-6 and row[9] == 0
        assert -1 <= row[7] <= 161 and -1 <= row[8] <= 97                   # the plate centre lands in the image


def test_pascal3d_view_skips_images_with_several_cars(tmp_path):
    import os
    import scipy.io as sio
    _pascal_set(str(tmp_path))
    seen = {}
    for f in sorted(os.listdir(str(tmp_path / 'car_imagenet_label'))):
        v = render.pascal3d_view(sio.loadmat(str(tmp_path / 'car_imagenet_label' / f)))
        seen[f] = v
        if v is not None:
            assert -0.2 < v[0] < 0.6 and 0 <= v[1] < 2 * math.pi + 1e-6 and len(v[2]) == 4 and v[2][2] > v[2][0]
    assert sum(v is None for v in seen.values()) == 1


@pytest.mark.gpu
def test_lp_generator_on_device(cuda, tmp_path):
    """LPGenerator.add: the host batch blended by yolo_composite_unit onto 0..1 images (RenderCar's output)."""
    import torch
    _fonts(str(tmp_path / 'fonts'))
    gen = render.LPGenerator(64, 96, str(tmp_path / 'fonts'), CAMERA)
    bg = torch.rand((4, 3, 64, 96)).to(cuda)
    import random
    np.random.seed(4); random.seed(4)
    img, lab = gen.add(bg, [45, 60, 45])
    np.random.seed(4); random.seed(4)
    fg, mask, lab_h = gen.add_host(4, 64, 96, [45, 60, 45])
    ref = np.clip(bg.cpu().numpy() * (1 - mask) + fg * mask, 0, 1)
    np.testing.assert_allclose(img.cpu().numpy(), ref, rtol=0, atol=1e-6)
    np.testing.assert_array_equal(lab.cpu().numpy(), lab_h)
    assert tuple(lab.shape) == (4, 1, 10) and bool((lab[:, 0, 0] == 1).all())
