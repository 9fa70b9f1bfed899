"""Deployment seam: the frozen-graph runner the reference builds with
`yolo_gluon.init_executor` (yolo_gluon.py:204-242: load an exported checkpoint, bind it for inference, optionally in
half precision / through TensorRT) and the `/YOLO/box` row the video node publishes (car/video_node.py:235-255).

On MI355X the "engine" is a `CarNet` with BatchNorm folded into per-channel scale/bias and the weights packed for
the MFMA kernels (`CarNet.prepare`), running bf16 activations: there is no separate runtime to build.  The exported
symbol JSON is MXNet-specific and is not needed -- the spec rebuilds the graph -- so only the `export-NNNN.params`
half of `net.export` (yolo_gluon.py:245-272: the gluon parameter names prefixed with `arg:` / `aux:`) is read and
written; parameters are matched by name (yolo_amd/mxparams.py).
"""
import math
import os

import numpy as np

from . import mxparams


class Executor(object):
    """What `init_executor` returns, with the call signature the video node uses:
    `net_out = executor.forward(is_train=False, data=nd_img)` (car/video_node.py:230)."""

    def __init__(self, net):
        self.net = net
        self.outputs = None

    def forward(self, is_train=False, data=None):
        if is_train:
            raise ValueError('the deployment executor is inference-only (grad_req="null", yolo_gluon.py:236)')
        out = self.net(data)
        self.outputs = out if isinstance(out, list) else list(out[0]) + list(out[1])
        return self.outputs


def export_params(net, export_folder, epoch=0, prefix='carnet0_'):
    """The parameter half of yolo_gluon.export: <folder>/export-%04d.params, what HybridBlock.export writes --
    collect_params() with every name prefixed by `arg:` (or `aux:` for the running statistics)."""
    os.makedirs(export_folder, exist_ok=True)
    path = os.path.join(export_folder, 'export-%04d.params' % epoch)
    mxparams.write_params(path, mxparams.to_gluon(net.graph, net.collect_params(), prefix, export=True))
    return path


def init_executor(export_folder, spec, size, device='cuda:0', step=0, dtype='bf16', tune='auto'):
    """yolo_gluon.init_executor(export_folder, size, ctx, use_tensor_rt, step, fp16) for MI355X: load
    <folder>/export-%04d.params, fold BN, pack the weights, pre-build the launch plan for a (1,3,H,W) input."""
    import torch
    from .net import CarNet, CarLPNet
    cls = CarLPNet if 'LP_slice_point' in spec else CarNet
    net = cls(spec, dtype=dtype, device=device, tune=tune)
    net.load_gluon_params(os.path.join(export_folder, 'export-%04d.params' % step))
    net.prepare()
    net(torch.zeros((1, 3, int(size[0]), int(size[1])), dtype=torch.float32, device=device))    # bind: build the plan
    return Executor(net)


_STEP = 360 // 24
_COS_OFFSET = np.array([math.cos(x * math.pi / 180) for x in range(0, 360, _STEP)])       # car/video_node.py:36-38
_SIN_OFFSET = np.array([math.sin(x * math.pi / 180) for x in range(0, 360, _STEP)])


def car_box_row(pred_car, depth_image=None):
    """Video.process (car/video_node.py:235-255): the row published on /YOLO/box from predict()'s (1, 6+24) output
    [score, y, x, h, w, rot, cls...]: element 5 becomes the azimuth = atan2 of the softmax-weighted mean direction of
    the 24 azimuth classes (15 degrees apart).  Returns a copy of the row."""
    row = np.array(pred_car[0], copy=True)
    x = row[-24:]
    prob = np.exp(x) / np.sum(np.exp(x), axis=0)
    c = sum(_COS_OFFSET * prob)
    s = sum(_SIN_OFFSET * prob)
    row[5] = math.atan2(s, c)
    return row
