"""hipGraph replay of the inference forward against eager launches (same box, same plan): does capturing the ~75 launches of
the D53 forward buy anything on the GPU side?    python tools/graph_probe.py [--dtype bf16] [--size 416] [--batch 32]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import plans
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--size', type=int, default=416)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--steps', type=int, default=100)
a = ap.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
net = CarNet(darknet53_spec(), dtype=a.dtype, device=dev, tune='measure').initialize(seed=1234)
net.load_tuning_state(plans.load(plans.DEFAULT)[0])
x = torch.rand((a.batch, 3, a.size, a.size), device=dev)
for _ in range(5):
    outs = net(x)
torch.cuda.synchronize()
ref = [o.clone() for o in outs]


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        net(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gouts = net(x)
g.replay()
torch.cuda.synchronize()
same = all(bool((p == q).all()) for p, q in zip(gouts, ref))
for rep in range(3):
    print('%s %dx%d bs %d: eager %.3f ms  graph %.3f ms  (bit-identical %s)' % (a.dtype, a.size, a.size, a.batch, timed(lambda: net(x), a.steps),
                                                                          timed(g.replay, a.steps), same), flush=True)
