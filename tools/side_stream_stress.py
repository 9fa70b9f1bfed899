#!/usr/bin/env python
"""Inference with the head tip/output convolutions on a side stream must be bit-identical to the single-stream pass
(same kernels, deterministic arithmetic): repeat and compare the merged logits bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec
dev = torch.device('cuda:0')
B, S, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.rand((B, 3, S, S), generator=torch.Generator().manual_seed(5)).to(dev)
nets = {}
for side in (False, True):
    net = CarNet(darknet53_spec(), dtype='bf16', device=dev, side_stream=side).initialize(seed=1234)
    net.prepare()
    nets[side] = net
ref = [o.clone() for o in nets[False](x)]
torch.cuda.synchronize()
bad = 0
for it in range(iters):
    for side in (False, True):
        outs = nets[side](x)
        torch.cuda.synchronize()
        for k, (o, r) in enumerate(zip(outs, ref)):
            n = int((o.view(torch.int32) != r.view(torch.int32)).sum())
            if n:
                bad += n
                print('iter %d side=%s scale %d: %d elements differ' % (it, side, k, n), flush=True)
print('side_stream_stress: %d differing elements over %d iterations' % (bad, iters))
