"""The oracle and the product's host code against outputs of THE REFERENCE ITSELF -- the few functions of it that run without mxnet,
executed in the build container by tests/golden/make_reference_vectors.py (which reads their definitions from /root/reference and
runs them on numpy; the vectors are data: inputs + the reference's outputs):

  * `predict_LP`, numpy branch (licence_plate/LP_detection.py:147-162)  -> oracle.detect.predict_LP
  * `np_sigmoid` / `np_inv_sigmoid` (yolo_modules/yolo_gluon.py:370-377)                -> oracle.detect.sigmoid, oracle.train.inv_sigmoid
  * `ProjectRectangle6D` (licence_plate_render/__init__.py:336-377, row f2)             -> oracle.render.project_plate,
                                                                                            yolo_amd.render.PlateCamera.corners / centre
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
