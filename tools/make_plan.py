#!/usr/bin/env python
"""Measure the kernel variants of every workload bench.py runs ON THIS BOX and write them as the committed launch plan:
    python tools/make_plan.py --commit <short hash> [--out profiles/plan.json] [--train-batches 64,32,128]
Workloads: D53 spec inference 416x416 bs 32 and 608x608 bs 64 (bf16), the fp32 parity path at 416x416 bs 32, and the training
step at 416x416 for the per-GPU batches of BASELINE configs[2] / configs[3] (64; 256 // N = 128, 32; N = 4 is 64 again).
bench.py loads the file by default (--tune plan): the bench line, the rocprofv3 kernel trace and the PMC passes under profiles/
then describe the same launches whatever box they ran on (yolo_amd/plans.py)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import plans
from yolo_amd.net import CarNet
from yolo_amd.train import Trainer
from yolo_amd.spec import darknet53_spec
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--out', default=plans.DEFAULT)
ap.add_argument('--commit', default=None)
ap.add_argument('--train-batches', default='64,32,128')
ap.add_argument('--no-f32', action='store_true')
ap.add_argument('--base', default=None, help='an existing plan file: its choices are kept, only shapes it does not hold are measured')
ap.add_argument('--remeasure', default='', help="comma-separated dtypes whose forward choices of --base are dropped and measured again (e.g. 'bf16x3' after new variants for it)")
a = ap.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
spec = darknet53_spec()
states, shapes, t0 = [], [], time.time()
base_workloads = []
if a.base:
    bstate, bmeta = plans.load(a.base)
    from yolo_amd import lib as L_
    drop = {{'f32': L_.F32, 'bf16': L_.BF16, 'f16': L_.F16, 'bf16x3': L_.BF16X3, 'f16x3': L_.F16X3}[d_] for d_ in a.remeasure.split(',') if d_}
    # (a forward conv key is (N, H, W, Cin, Cout, ksize, stride, out_f32, residual, dtype, ...): plans of net.py _measure_algo)
    bstate['algo'] = {k: v for k, v in bstate['algo'].items() if not (isinstance(k[0], int) and len(k) >= 10 and k[9] in drop)}
    states.append(bstate)
    base_workloads = list(bmeta.get('workloads', []))

net = CarNet(spec, dtype='bf16', device=dev, tune='measure').initialize(seed=1234)
if states:
    net.load_tuning_state(plans.merge(*states))
net.prepare()
for B, S in ((32, 416), (64, 608)):
    net.plan_signature(B, S, S)
    shapes.append('infer bf16 %dx%d bs %d' % (S, S, B))
    print(shapes[-1], '%.0f s' % (time.time() - t0), flush=True)
states.append(net.tuning_state())
del net
torch.cuda.empty_cache()
if not a.no_f32:
    net = CarNet(spec, dtype='f32', device=dev, tune='measure').initialize(seed=1234)
    net.load_tuning_state(plans.merge(*states))
    net.prepare()
    net.plan_signature(32, 416, 416)
    states.append(net.tuning_state())
    shapes.append('infer f32 416x416 bs 32')
    print(shapes[-1], '%.0f s' % (time.time() - t0), flush=True)
    del net
    torch.cuda.empty_cache()
for dt in ('f16', 'bf16x3', 'f16x3'):     # (bf16x3: the split bf16 parity path, bench.py's `parity_path` key and --dtype bf16x3; f16x3: its IEEE-half sibling)
    net = CarNet(spec, dtype=dt, device=dev, tune='measure').initialize(seed=1234)
    net.load_tuning_state(plans.merge(*states))
    net.prepare()
    for B, S in ((32, 416), (64, 608)):
        net.plan_signature(B, S, S)
        shapes.append('infer %s %dx%d bs %d' % (dt, S, S, B))
        print(shapes[-1], '%.0f s' % (time.time() - t0), flush=True)
    states.append(net.tuning_state())
    del net
    torch.cuda.empty_cache()
for B in [int(v) for v in a.train_batches.split(',') if v]:
    net = CarNet(spec, dtype='bf16', device=dev, tune='measure').initialize(seed=1234)
    tr = Trainer(net, (416, 416))
    tr.load_tuning_state(plans.merge(*states))            # shapes already measured keep their choice
    x = torch.rand((B, 3, 416, 416), device=dev)
    lab = torch.from_numpy(bench.synthetic_labels(B, 3)).to(dev)
    states.append(tr.tune(x, lab))
    shapes.append('train bf16 416x416 bs %d' % B)
    print(shapes[-1], '%.0f s' % (time.time() - t0), flush=True)
    del tr, net, x
    torch.cuda.empty_cache()
state = plans.merge(*states)
meta = {'commit': a.commit, 'workloads': shapes, 'base': (a.base and os.path.basename(a.base)), 'base_workloads': base_workloads, 'device': torch.cuda.get_device_name(0), 'made': time.strftime('%Y-%m-%d %H:%M:%S'),
        'choices': {s: len(state[s]) for s in plans.SECTIONS}}
plans.save(a.out, state, meta)
print('wrote %s  md5 %s  %s' % (a.out, plans.md5(state), meta['choices']))
