#!/usr/bin/env python
"""Golden vectors produced by THE REFERENCE ITSELF, run in the build container (no GPU, no mxnet):

    python tests/golden/make_reference_vectors.py            -> tests/golden/reference_vectors.npz

The reference's modules import mxnet at the top and three of its drivers are Python 2, so none of them can be imported whole.
The functions used here do not touch mxnet on the path exercised:

  * `LicencePlateDetectioin.predict_LP`, numpy branch (`use_np`)        licence_plate/LP_detection.py:147-162   
  * `np_sigmoid`, `np_inv_sigmoid`                                        yolo_modules/yolo_gluon.py:370-377       (used by a21 / a15)
  * `ProjectRectangle6D.__call__` / `.projection_matrix`                  yolo_modules/licence_plate_render/__init__.py:336-377  (row f2)
  * the camera calibration the projection reads                          camera_parameter/C310_4.yaml
  * `LPGenerator.draw_LP` (PIL + numpy.random) with the statements of `LPGenerator.__init__` that set up what it uses (plate sizes,
    glyph columns, the glyph / dot images and their resizing), on a SYNTHETIC glyph set (tests/test_render.py:_fonts -- the
    reference's glyph PNGs stay with the reference) and `yolo_cv._color`           licence_plate_render/__init__.py:23-40,60-77,
                                                                                  yolo_modules/yolo_cv.py:11-20   (row f2)
  * `YOLO._init_step`, `YOLO._init_area` (the anchor grid's strides and cell counts)      car/YOLO.py:112-121 (row a10)
  * `RadarProb.cls2ang` / `_numpy_softmax` (azimuth of the published box row)           yolo_modules/yolo_cv.py:85-95, 234-236 (row f4)
  * `yolo_cv.PILImageEnhance` (random rotate / blur / noise) and `RenderCar._resize`      yolo_modules/yolo_cv.py:97-157, car/render_car.py:379-407 (row f2)

Their DEFINITIONS are read from the reference's files where they lie under /root/reference -- located with `ast` (or, in the Python-2
file, by their `def` line and indentation) -- and executed with their real dependencies (numpy, math).  Nothing of mxnet is stubbed
and no reference source is written into this repository: the output holds inputs and the reference's outputs only (data), and the
reference does not travel to the GPU box.  tests/test_reference_vectors.py holds the oracle AND the product's host code to these vectors;
tests/test_gpu_golden.py holds the HIP `predict_LP` kernel to them.
"""
import ast
import math
import os
import textwrap
import types

import numpy as np
import yaml

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_vectors.npz')


def _module_function(path, name, cls=None):
    """Source of a module-level function (or a method of class `cls`) of a Python-3-parsable reference file."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    scope = tree.body
    if cls is not None:
        scope = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in scope if isinstance(n, ast.FunctionDef) and n.name == name)
    return textwrap.dedent('\n'.join(src.split('\n')[node.lineno - 1:node.end_lineno]))


def _method_by_lines(path, name):
    """Source of a method of a file that does not parse as Python 3 as a whole (`exec "..."` statements elsewhere in it): from its
    `def` line to the next line of the same or smaller indentation."""
    lines = open(os.path.join(REF, path)).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.lstrip().startswith('def %s(' % name))
    ind = len(lines[start]) - len(lines[start].lstrip())
    end = start + 1
    while end < len(lines) and (not lines[end].strip() or len(lines[end]) - len(lines[end].lstrip()) > ind):
        end += 1
    return textwrap.dedent('\n'.join(lines[start:end]))


def _exec(src, ns):
    exec(compile(ast.parse(src), '<reference>', 'exec'), ns)
    return ns


def main(out_path=OUT):
    out = {}
    # ---- np_sigmoid / np_inv_sigmoid (the reference's module calls numpy `numpy`) ----------------------------------------------
    g = {'numpy': np}
    _exec(_module_function('yolo_modules/yolo_gluon.py', 'np_sigmoid'), g)
    _exec(_module_function('yolo_modules/yolo_gluon.py', 'np_inv_sigmoid'), g)
    rng = np.random.default_rng(20261002)
    x = np.concatenate([np.linspace(-20, 20, 81), rng.standard_normal(64) * 4]).astype(np.float32)
    p = np.concatenate([np.linspace(1e-4, 0.9999, 41), rng.uniform(0.01, 0.99, 32)]).astype(np.float32)
    out['sigmoid_x'], out['sigmoid_y'] = x, g['np_sigmoid'](x)
    out['inv_sigmoid_p'], out['inv_sigmoid_y'] = p, g['np_inv_sigmoid'](p)
    # ---- predict_LP, numpy branch ------------------------------------------------------------------------------------------------
    ns = {'np': np, 'math': math, 'yolo_gluon': types.SimpleNamespace(np_sigmoid=g['np_sigmoid'])}
    _exec(_method_by_lines('licence_plate/LP_detection.py', 'predict_LP'), ns)
    spec = yaml.safe_load(open(os.path.join(REF, 'licence_plate/v1/spec.yaml')))
    LP_slice_point = [int(v) for v in spec['LP_slice_point']]
    cases_in, cases_out, cases_rmax = [], [], []
    for k, (h, w) in enumerate([(10, 16), (13, 13), (20, 32), (1, 1), (26, 26), (10, 16), (5, 7), (10, 16)]):
        r_max = [float(v) for v in (spec['LP_r_max'] if k % 2 == 0 else rng.uniform(10, 80, 3))]
        self_ = types.SimpleNamespace(LP_slice_point=LP_slice_point, LP_r_max=r_max)
        b = (rng.standard_normal((1, LP_slice_point[-1], h, w)) * 2).astype(np.float32)
        if k == 5:
            b[0, 0] = 0.25                                   # every cell ties: the first index wins
        cases_in.append(b.copy())                            # (predict_LP writes into a view of its input)
        cases_out.append(np.array(ns['predict_LP'](self_, b), np.float32))
        cases_rmax.append(np.array(r_max, np.float64))
    for k, (a, b_, c) in enumerate(zip(cases_in, cases_out, cases_rmax)):
        out['lp_in_%d' % k], out['lp_out_%d' % k], out['lp_rmax_%d' % k] = a, b_, c
    out['lp_cases'] = np.array(len(cases_in))
    out['lp_slice_point'] = np.array(LP_slice_point)
    # ---- ProjectRectangle6D: the plate's corners in camera pixels ------------------------------------------------------------------
    cam = yaml.safe_load(open(os.path.join(REF, 'camera_parameter/C310_4.yaml')))
    P = cam['projection_matrix']['data']
    pn = {'np': np, 'math': math}
    body = {}
    _exec(_module_function('yolo_modules/licence_plate_render/__init__.py', '__call__', cls='ProjectRectangle6D'), body.setdefault('ns', dict(pn)))
    _exec(_module_function('yolo_modules/licence_plate_render/__init__.py', 'projection_matrix', cls='ProjectRectangle6D'), body['ns'])
    Proj = type('ProjectRectangle6D', (), {'__call__': body['ns']['__call__'], 'projection_matrix': body['ns']['projection_matrix']})
    proj = Proj()
    # (what the reference's __init__ reads from the yaml, :279-285)
    proj.camera_w, proj.camera_h = cam['image_width'], cam['image_height']
    proj.fx, proj.fy, proj.cx, proj.cy = P[0], P[5], P[2], P[6]
    poses = np.stack([np.concatenate([rng.uniform(-1500, 1500, 2), rng.uniform(1000, 8000, 1), np.deg2rad(rng.uniform(-60, 60, 3))])
                      for _ in range(64)])
    poses[0] = [0, 0, 3000, 0, 0, 0]
    out['proj_poses'] = poses
    out['proj_points'] = np.stack([proj(list(q)) for q in poses])
    out['proj_camera'] = np.array([cam['image_width'], cam['image_height'], P[0], P[5], P[2], P[6]], np.float64)
    # ---- LPGenerator.draw_LP: the random draw sequence and the paste geometry of a plate -----------------------------------------------
    import sys
    import tempfile
    import PIL.Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    from test_render import _fonts                       # the synthetic glyph set the repo's own render tests use
    lsrc = open(os.path.join(REF, 'yolo_modules/licence_plate_render/__init__.py')).read()
    ltree = ast.parse(lsrc)
    lcls = next(n for n in ltree.body if isinstance(n, ast.ClassDef) and n.name == 'LPGenerator')
    init = next(n for n in lcls.body if isinstance(n, ast.FunctionDef) and n.name == '__init__')
    seg = lambda n: textwrap.dedent('\n'.join(lsrc.split('\n')[n.lineno - 1:n.end_lineno]))
    # the statements of __init__ that prepare what draw_LP reads; the ones that need mxnet / yolo_cv objects / the module's own
    # directory are left out (fonts_dir is bound to the synthetic set instead)
    keep = [seg(n) for n in init.body if not any(w in seg(n) for w in ('mxnet', 'yolo_cv', 'ProjectRectangle6D', 'module_dir'))]
    csrc = open(os.path.join(REF, 'yolo_modules/yolo_cv.py')).read()
    cnode = next(n for n in ast.parse(csrc).body if isinstance(n, ast.Assign) and getattr(n.targets[0], 'id', '') == '_color')
    colors = ast.literal_eval(cnode.value)
    with tempfile.TemporaryDirectory() as tmp:
        fonts_dir = os.path.join(tmp, 'fonts')
        _fonts(fonts_dir)
        gen = types.SimpleNamespace()
        ns2 = {'np': np, 'PIL': PIL, 'os': os, 'self': gen, 'fonts_dir': fonts_dir, 'img_h': 96, 'img_w': 160, 'class_index': 1,
               'yolo_cv': types.SimpleNamespace(_color=colors)}
        for stmt in keep:
            _exec(stmt, ns2)
        _exec(_module_function('yolo_modules/licence_plate_render/__init__.py', 'draw_LP', cls='LPGenerator'), ns2)
        for k, seed in enumerate((0, 8, 2026)):
            np.random.seed(seed)
            plate, lp_type, label = ns2['draw_LP'](gen)
            out['plate_seed_%d' % k] = np.array(seed)
            out['plate_rgba_%d' % k] = np.asarray(plate).copy()
            out['plate_label_%d' % k] = np.array(label, np.float64)
            out['plate_type_%d' % k] = np.array(lp_type)
    out['plate_cases'] = np.array(3)
    # ---- yolo_cv.PILImageEnhance (rotate / blur / noise: PIL + numpy.random; the module imports cv2, the class does not use it) and
    #      RenderCar._resize, on a synthetic RGBA sprite ------------------------------------------------------------------------------
    import PIL.ImageFilter
    cnode = next(n for n in ast.parse(csrc).body if isinstance(n, ast.ClassDef) and n.name == 'PILImageEnhance')
    ens = {'np': np, 'PIL': PIL}
    _exec(textwrap.dedent('\n'.join(csrc.split('\n')[cnode.lineno - 1:cnode.end_lineno])), ens)
    rns = {'np': np, 'PIL': PIL}
    _exec(_module_function('car/render_car.py', '_resize', cls='RenderCar'), rns)
    yy, xx = np.mgrid[0:48, 0:80]
    sprite = np.zeros((48, 80, 4), np.uint8)
    sprite[..., 0], sprite[..., 1], sprite[..., 2] = (xx * 3) % 256, (yy * 5) % 256, ((xx + yy) * 2) % 256
    sprite[8:40, 10:70, 3] = 255
    out['sprite'] = sprite
    enh = [(30.0, 0.3, 0.0), (0.0, 1.0, 10.0), (0.0, 1.0, 5.0), (30.0, 0.0, 0.0), (15.0, 0.5, 3.0)]      # (R, G, noise_var): RenderCar :43-44, LPGenerator :46-47 / :124
    for k, (R_, G_, nv) in enumerate(enh):
        np.random.seed(100 + k)
        img, r = ens['PILImageEnhance'](M=0., N=0., R=R_, G=G_, noise_var=nv)(PIL.Image.fromarray(sprite))
        out['enh_args_%d' % k], out['enh_img_%d' % k], out['enh_r_%d' % k] = np.array([R_, G_, nv]), np.asarray(img).copy(), np.array(r, np.float64)
    out['enh_cases'] = np.array(len(enh))
    for k, (lo, hi, r1) in enumerate([(0.2, 1.0, 1.0), (0.5, 0.6, 0.8), (1.0, 2.0, 1.3)]):
        np.random.seed(200 + k)
        resize, rw, rh, img = rns['_resize'](None, PIL.Image.fromarray(sprite), lo, hi, r1)
        out['rsz_args_%d' % k], out['rsz_out_%d' % k], out['rsz_img_%d' % k] = np.array([lo, hi, r1]), np.array([resize, rw, rh], np.float64), np.asarray(img).copy()
    out['rsz_cases'] = np.array(3)
    # ---- YOLO._init_step / _init_area (car/YOLO.py:112-121, SURVEY row a10: plain Python arithmetic) -----------------------------------
    gns = {}
    _exec(_method_by_lines('car/YOLO.py', '_init_step'), gns)
    _exec(_method_by_lines('car/YOLO.py', '_init_area'), gns)
    car_v1 = yaml.safe_load(open(os.path.join(REF, 'car/v1/spec.yaml')))
    test_y = yaml.safe_load(open(os.path.join(REF, 'yolo_modules/test.yaml')))
    grid_cases = [(car_v1['layers'], car_v1['all_anchors'], (320, 512)), (test_y['layers'], test_y['all_anchors'], (192, 256)),
                  ([1, 2, 8, 8, 4], car_v1['all_anchors'], (416, 416)), ([1, 2, 8, 8, 4], car_v1['all_anchors'], (608, 608)),
                  ([1, 1, 2, 1, 1], car_v1['all_anchors'], (64, 96))]
    for k, (layers, anchors, size) in enumerate(grid_cases):
        obj = types.SimpleNamespace(layers=layers, all_anchors=anchors, size=list(size))
        gns['_init_step'](obj)
        gns['_init_area'](obj)
        out['grid_layers_%d' % k], out['grid_nscale_%d' % k], out['grid_size_%d' % k] = np.array(layers), np.array(len(anchors)), np.array(size)
        out['grid_steps_%d' % k], out['grid_area_%d' % k] = np.array(obj.steps), np.array(obj.area)
    out['grid_cases'] = np.array(len(grid_cases))
    # ---- RadarProb.cls2ang + _numpy_softmax (yolo_cv.py:85-95, 234-236): the azimuth of the /YOLO/box row = atan2 of the softmax-weighted
    #      mean direction of the 24 azimuth classes -- the arithmetic car/video_node.py:244-251 repeats inline (row f4).  The direction
    #      tables are INPUT here (cos / sin of k * 15 degrees): the reference builds them with Python-2 integer division (:24-26)
    ans = {'np': np, 'math': math}
    _exec(_module_function('yolo_modules/yolo_cv.py', '_numpy_softmax'), ans)
    _exec(_module_function('yolo_modules/yolo_cv.py', 'cls2ang', cls='RadarProb'), ans)
    radar = types.SimpleNamespace(cos_offset=np.array([math.cos(k * 15 * math.pi / 180) for k in range(24)]),
                                  sin_offset=np.array([math.sin(k * 15 * math.pi / 180) for k in range(24)]))
    logits = (rng.standard_normal((16, 24)) * 3).astype(np.float32)
    logits[0] = 0.0
    logits[1, 5] = 30.0
    res = [ans['cls2ang'](radar, 0.75, logits[k].copy()) for k in range(len(logits))]
    out['azi_logits'] = logits
    out['azi_angle'] = np.array([r_[0] for r_ in res], np.float64)
    out['azi_radius'] = np.array([r_[1] for r_ in res], np.float64)
    np.savez_compressed(out_path, **out)
    print('wrote %s: %d arrays, %d bytes' % (out_path, len(out), os.path.getsize(out_path)))


if __name__ == '__main__':
    main()
