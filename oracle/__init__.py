"""CPU oracle for the YOLOv3 hot path of n8886919/YOLO.

TEST INFRASTRUCTURE ONLY.  Nothing under ``yolo_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and there only as the checker / the thing timed as a CPU baseline.

PARITY UNPINNED: the reference has no tests, golden vectors or known-answer
fixtures, and its arithmetic lives in mxnet / gluoncv
0.4.0b20181129 (requirements.txt:1,9), neither of which is installed or
installable here, so the reference itself cannot be run to pin this oracle.
What pins it instead: two independent restatements
(numpy fp64 direct convolution in ``forward.forward_numpy64`` and torch-CPU
fp32 in ``forward.forward_torch``) that must agree, analytic known answers
(zero logits -> score 0.5, boxes = anchors at cell centres; IoU(self) = 1), and
the structural cross-check against the shapes the reference author recorded
in comments (car/YOLO.py:135, :661-662).

PINNED, since round 6, are the few rows whose reference code runs without mxnet:
``detect.predict_LP`` (licence_plate/LP_detection.py:147-162, its numpy branch),
``detect.init_steps`` / ``init_area`` (car/YOLO.py:112-121),
``detect.sigmoid`` / ``train.inv_sigmoid`` (yolo_gluon.py:370-377),
``render.project_plate`` (ProjectRectangle6D, licence_plate_render/__init__.py:336-377),
``render.draw_plate`` (LPGenerator.draw_LP, :60-77) and ``render.enhance``
(yolo_cv.PILImageEnhance, yolo_cv.py:97-157)
(and, product only, the box row's azimuth: RadarProb.cls2ang, yolo_cv.py:85-95)
are held to outputs of the reference's OWN functions, executed in the build
container by tests/golden/make_reference_vectors.py (it reads their definitions
from /root/reference and runs them on numpy; the committed vectors are data) --
tests/test_reference_vectors.py.  Everything that needs mxnet / gluoncv -- the
network, decode, losses, Adam -- stays unpinned.
"""
