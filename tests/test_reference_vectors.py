"""The oracle and the product's host code against outputs of THE REFERENCE ITSELF -- the few functions of it that run without mxnet,
executed in the build container by tests/golden/make_reference_vectors.py (which reads their definitions from /root/reference and
runs them on numpy; the vectors are data: inputs + the reference's outputs):

  * `predict_LP`, numpy branch (licence_plate/LP_detection.py:147-162)  -> oracle.detect.predict_LP
  * `np_sigmoid` / `np_inv_sigmoid` (yolo_modules/yolo_gluon.py:370-377)                -> oracle.detect.sigmoid, oracle.train.inv_sigmoid
  * `ProjectRectangle6D` (licence_plate_render/__init__.py:336-377, row f2)             -> oracle.render.project_plate,
                                                                                            yolo_amd.render.PlateCamera.corners / centre
  * `LPGenerator.draw_LP` (:60-77), `yolo_cv.PILImageEnhance` (yolo_cv.py:97-157),
    `RenderCar._resize` (render_car.py:379-407), all row f2                              -> oracle.render.draw_plate / enhance,
                                                                                            yolo_amd.render.LPGenerator / RenderCar
  * `YOLO._init_step` / `_init_area` (car/YOLO.py:112-121, row a10)                     -> oracle.detect.init_steps / init_area,
                                                                                            NetGraph.steps, make_grid
These rows of the oracle are PINNED; everything else stays "parity unpinned" (oracle/__init__.py).  The HIP `predict_LP` kernel is
held to the same vectors by tests/test_gpu_golden.py::test_predict_lp_vs_the_reference_itself.
"""
import os

import numpy as np

from oracle import detect as od, train as ot, render as orr

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors.npz'))


def test_sigmoid_and_inverse_match_the_reference():
    y = od.sigmoid(G['sigmoid_x'])
    np.testing.assert_allclose(y, G['sigmoid_y'], rtol=2e-7, atol=0)
    inv = np.array([ot.inv_sigmoid(np.float32(p)) for p in G['inv_sigmoid_p']], np.float32)
    np.testing.assert_allclose(inv, G['inv_sigmoid_y'], rtol=1e-6, atol=1e-6)


def test_predict_LP_matches_the_reference():
    """Every case: the same best cell (first index among ties, case 5) and the same pose row, to a float32 ulp or two (the reference's
    scalar arithmetic promotes through Python floats differently under its own numpy 1.x and the numpy 2.x that produced the vectors)."""
    assert list(G['lp_slice_point']) == [1, 3, 4, 7, 10]
    for k in range(int(G['lp_cases'])):
        x, want, r_max = G['lp_in_%d' % k], G['lp_out_%d' % k], list(G['lp_rmax_%d' % k])
        got, best = od.predict_LP(x, r_max)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7, err_msg='case %d' % k)
        flat = x.transpose(0, 2, 3, 1)[0].reshape(-1, x.shape[1])
        assert np.array_equal(flat[best][7:], want[7:])                    # the row of the chosen cell: untouched class channels
    assert int(np.argmax(G['lp_in_5'][0, 0].reshape(-1))) == 0             # the tie case really is a tie over all cells


def test_plate_projection_matches_the_reference():
    from yolo_amd.render import PlateCamera
    w, h, fx, fy, cx, cy = [float(v) for v in G['proj_camera']]
    cam = PlateCamera({'image_width': int(w), 'image_height': int(h),
                       'projection_matrix': {'data': [fx, 0.0, cx, 0.0, 0.0, fy, cy, 0.0, 0.0, 0.0, 1.0, 0.0]}})
    for pose, want in zip(G['proj_poses'], G['proj_points']):
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(orr.project_plate(list(pose), dict(fx=fx, fy=fy, cx=cx, cy=cy)), want, rtol=0, atol=2e-6 * scale)
        np.testing.assert_allclose(cam.corners(list(pose)), want, rtol=0, atol=2e-5 * scale)
    # the frontal pose at 3 m: a 399 x 168 mm plate, centred on the optical axis
    p0 = G['proj_points'][0]
    assert abs((p0[0, 0] - p0[1, 0]) - fx * 399.0 / 3000.0) < 1e-3 and abs((p0[0, 1] - p0[3, 1]) - fy * 168.0 / 3000.0) < 1e-3


def test_plate_drawing_matches_the_reference(tmp_path):
    """`LPGenerator.draw_LP` (licence_plate_render/__init__.py:60-77) run from the reference with numpy's global seed set, on the
    synthetic glyph set: the product's `LPGenerator.draw_LP` and the oracle's `draw_plate` draw the same letters and digits (the same
    `np.random` sequence: three of 10..33, four of 0..8 with 4 -> 9), paste them at the same columns -- the same RGBA plate, pixel
    for pixel -- and return the same glyph labels."""
    from PIL import Image
    from test_render import _fonts, CAMERA
    from yolo_amd import render
    _fonts(str(tmp_path / 'fonts'))
    gen = render.LPGenerator(96, 160, str(tmp_path / 'fonts'), CAMERA, augment=False)
    font = [Image.open(str(tmp_path / 'fonts' / ('%d.png' % k))).resize((45, 90), Image.BILINEAR) for k in range(34)]
    dot = Image.open(str(tmp_path / 'fonts' / '34.png')).resize((10, 70), Image.BILINEAR)
    for k in range(int(G['plate_cases'])):
        seed = int(G['plate_seed_%d' % k])
        np.random.seed(seed)
        plate, lp_type, glyphs = gen.draw_LP()
        assert lp_type == int(G['plate_type_%d' % k])
        assert np.array_equal(np.asarray(plate), G['plate_rgba_%d' % k]), 'seed %d: the plate differs from the reference\'s' % seed
        np.testing.assert_allclose(np.array(glyphs, np.float64), G['plate_label_%d' % k], rtol=1e-12)
        np.random.seed(seed)
        assert np.array_equal(np.asarray(orr.draw_plate(font, dot)), G['plate_rgba_%d' % k])


def test_enhance_and_resize_match_the_reference():
    """`yolo_cv.PILImageEnhance.__call__` (random rotate by U(-R, R) with expand, Gaussian blur of radius rand() * G, N(0, noise_var)
    noise; yolo_cv.py:97-157) and `RenderCar._resize` (render_car.py:379-407), run from the reference under numpy's global seed on a
    synthetic RGBA sprite: the oracle's `enhance` and the product's `RenderCar._enhance` / `_resize` consume the random stream in the
    same order and return the same pixels."""
    from PIL import Image
    from yolo_amd import render
    sprite = Image.fromarray(G['sprite'])
    for k in range(int(G['enh_cases'])):
        R_, G_, nv = [float(v) for v in G['enh_args_%d' % k]]
        np.random.seed(100 + k)
        img, r = orr.enhance(sprite, R=R_, G=G_, noise_var=nv)
        assert np.array_equal(np.asarray(img), G['enh_img_%d' % k]) and abs(r - float(G['enh_r_%d' % k])) < 1e-15, 'oracle, case %d' % k
        if nv == 0.0:                                      # (RenderCar's enhancer: noise_var = 0, render_car.py:43-44)
            rc = object.__new__(render.RenderCar)
            rc.R, rc.G = R_, G_
            np.random.seed(100 + k)
            img, r = rc._enhance(sprite)
            assert np.array_equal(np.asarray(img), G['enh_img_%d' % k]) and abs(r - float(G['enh_r_%d' % k])) < 1e-15, 'product, case %d' % k
    rc = object.__new__(render.RenderCar)
    for k in range(int(G['rsz_cases'])):
        lo, hi, r1 = [float(v) for v in G['rsz_args_%d' % k]]
        np.random.seed(200 + k)
        resize, rw, rh, img = rc._resize(sprite, lo, hi, r1)
        np.testing.assert_allclose([resize, rw, rh], G['rsz_out_%d' % k], rtol=1e-15)
        assert np.array_equal(np.asarray(img), G['rsz_img_%d' % k])


def test_grid_steps_and_areas_match_the_reference():
    """`YOLO._init_step` / `_init_area` (car/YOLO.py:112-121, row a10) run from the reference for its own car/v1 and test.yaml specs and
    the D53 / micro specs: the oracle's `init_steps` / `init_area`, the product's `NetGraph.steps()` and the grid descriptor the HIP
    decode reads (`make_grid`: cells per scale) give the same strides and cell counts."""
    from yolo_amd.spec import NetGraph
    from yolo_amd.detect import make_grid
    anchors3 = [[[0.1, 0.1]] * 3] * 3
    for k in range(int(G['grid_cases'])):
        layers, nscale, size = [int(v) for v in G['grid_layers_%d' % k]], int(G['grid_nscale_%d' % k]), tuple(int(v) for v in G['grid_size_%d' % k])
        steps, area = [int(v) for v in G['grid_steps_%d' % k]], [int(v) for v in G['grid_area_%d' % k]]
        anchors = anchors3[:nscale]
        assert od.init_steps(layers, anchors) == steps
        assert od.init_area(size, steps) == area
        spec = dict(layers=layers, channels=[8 * 2 ** i for i in range(len(layers) + 1)], all_anchors=anchors, slice_point=[1, 3, 5, 6, 30])
        assert NetGraph(spec).steps() == steps
        g, nbox = make_grid(anchors, size, steps)
        assert [g.gh[i] * g.gw[i] for i in range(nscale)] == area and nbox == 3 * sum(area) and [g.step[i] for i in range(nscale)] == steps


def test_box_row_azimuth_matches_the_reference():
    """`RadarProb.cls2ang` (yolo_cv.py:85-95) run from the reference: the azimuth `deploy.car_box_row` writes into element 5 of the
    published /YOLO/box row (car/video_node.py:244-251 repeats the same arithmetic inline) is the reference's angle."""
    from yolo_amd.deploy import car_box_row
    for logits, want in zip(G['azi_logits'], G['azi_angle']):
        pred = np.zeros((1, 30), np.float32)
        pred[0, 6:] = logits
        got = car_box_row(pred)[5]
        assert abs(float(got) - float(want)) < 2e-6 or abs(abs(float(got) - float(want)) - 2 * np.pi) < 2e-6, (got, want)
    assert abs(G['azi_radius'][0] - 0.0) < 1e-9                           # uniform classes: no direction (radius 0)


def test_the_vectors_regenerate_from_the_reference(tmp_path):
    """Where the reference is present (the build container; it never travels to the GPU box), running the generator again -- i.e. the
    reference's own functions, read from /root/reference and executed -- reproduces the committed vectors array for array."""
    import importlib.util
    import pytest
    if not os.path.isdir('/root/reference/yolo_modules'):
        pytest.skip('the reference is not present here')
    spec = importlib.util.spec_from_file_location('make_reference_vectors', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                                                                                          'make_reference_vectors.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / 'again.npz')
    mod.main(out)
    again = np.load(out)
    assert sorted(again.files) == sorted(G.files)
    for k in G.files:
        assert np.array_equal(again[k], G[k]), k
