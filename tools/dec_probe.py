"""Times yolo_decode_scores alone at 608x608 batch 64 (22743 boxes x 30 values per image; 490 MB in and out)."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from yolo_amd.detect import Detector
from oracle import graph as og, detect as od
spec = og.spec_d53()
size = (608, 608)
steps = od.init_steps(spec['layers'], spec['all_anchors'])
det = Detector(spec, size, steps, device='cuda:0')
B = 64
print(det.nbox, det.C)
m = torch.randn(B, det.nbox // det.grid.A, det.grid.A, det.C, device='cuda:0')
if True:
    for _ in range(3): det.decode_scores(m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): det.decode_scores(m)
    e1.record(); torch.cuda.synchronize()
    print('decode_scores us', e0.elapsed_time(e1) / 20 * 1000)
