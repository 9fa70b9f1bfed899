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
"""
