#!/usr/bin/env python
"""Golden vectors produced by THE REFERENCE ITSELF, run in the build container (no GPU, no mxnet):

    python tests/golden/make_reference_vectors.py            -> tests/golden/reference_vectors.npz

The reference's modules import mxnet at the top and three of its drivers are Python 2, so none of them can be imported whole.
The functions used here do not touch mxnet on the path exercised:

  * `LicencePlateDetectioin.predict_LP`, numpy branch (`use_np`)        licence_plate/LP_detection.py:147-162   
  * `np_sigmoid`, `np_inv_sigmoid`                                        yolo_modules/yolo_gluon.py:370-377       (used by a21 / a15)
  * `ProjectRectangle6D.__call__` / `.projection_matrix`                  yolo_modules/licence_plate_render/__init__.py:336-377  (row f2)
  * the camera calibration the projection reads                          camera_parameter/C310_4.yaml

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


def main():
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
    np.savez_compressed(OUT, **out)
    print('wrote %s: %d arrays, %d bytes' % (OUT, len(out), os.path.getsize(OUT)))


if __name__ == '__main__':
    main()
