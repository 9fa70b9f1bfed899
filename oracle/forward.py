"""Oracle: network forward, two independent restatements (test infrastructure only).

Restates CarNet.hybrid_forward (car/utils.py:68-95) over the graph of oracle/graph.py:
stages -> routes (last num_pyramid stage outputs) -> heads deep->shallow with
transition, 2x nearest up-sample (gluoncv _upsample: repeat on W then H) and
concat([upsampled, route], dim=1) (car/utils.py:92-93); YOLOOutput (basic_yolo.py:98-103):
1x1 conv + bias -> transpose(0,2,3,1) -> reshape(B,-1,A,C); returns fine->coarse
(car/utils.py:95).

forward_torch   : torch CPU fp32 (F.conv2d / batch_norm / leaky_relu)
forward_numpy64 : numpy fp64, convolution written as explicit tap loops + einsum
forward_torch_f16sim  : the same with IEEE half (the reference's use_fp16)
forward_torch_bf16sim : fp32 math with operands rounded to bf16 at the points where the
                  HIP bf16 path rounds (weights; every conv input) -- the "rounding-aware"
                  oracle for the bf16 MFMA path (SURVEY.md section 7 hard parts).

PARITY UNPINNED (see oracle/__init__.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .graph import BN_EPS, LEAKY


# ----------------------------------------------------------------------------- torch fp32
def _t(P, name):
    v = P[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


_SIM_DT = [torch.bfloat16]          # the 2-byte type the rounding simulation rounds to (forward_torch_f16sim swaps it)


def _bf16_round(x):
    return x.to(_SIM_DT[0]).to(torch.float32)


def _conv_bn_act_torch(P, c, x, training=False, bn_stats=None, sim_bf16=False, round_out=True):
    """sim_bf16: weights and conv input rounded to bf16, fp32 accumulate, and (round_out) the
    activation rounded to bf16 as the HIP path stores it.  The conv feeding a residual add is
    NOT rounded on its own: the HIP epilogue adds the residual in fp32 and rounds once."""
    w = _t(P, c['name'] + '.weight')
    if sim_bf16:
        w = _bf16_round(w)
        x = _bf16_round(x)
    y = F.conv2d(x, w, None, stride=c['stride'], padding=c['pad'])
    if not c['bn']:
        return y + _t(P, c['name'] + '.bias').view(1, -1, 1, 1)
    g, b = _t(P, c['name'] + '.gamma'), _t(P, c['name'] + '.beta')
    if training and sim_bf16:
        y = _bf16_round(y)          # the bf16 training path stores the pre-BN conv output in bf16
    if training:
        # Gluon BatchNorm train mode: biased batch variance
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
        if bn_stats is not None:
            bn_stats[c['name']] = (mean.detach().clone(), var.detach().clone())
    else:
        mean, var = _t(P, c['name'] + '.running_mean'), _t(P, c['name'] + '.running_var')
    y = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS) * g.view(1, -1, 1, 1) \
        + b.view(1, -1, 1, 1)
    y = F.leaky_relu(y, LEAKY)
    return _bf16_round(y) if (sim_bf16 and round_out) else y


def _upsample2(x):
    # gluoncv _upsample(x, stride=2): x.repeat(axis=-1, 2).repeat(axis=-2, 2)
    return x.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2)


def _yolo_output(y, num_anchors, per_anchor):
    # basic_yolo.py:102-103
    B = y.shape[0]
    return y.permute(0, 2, 3, 1).reshape(B, -1, num_anchors, per_anchor)


def forward_torch(g, P, x, training=False, bn_stats=None, sim_bf16=False, taps=None):
    """x: (B,3,H,W) float32 torch tensor or ndarray.  Returns list of 3 tensors fine->coarse,
    each (B, H_i*W_i, A, C) -- and, for a spec with an LP branch (CarLPNet), the tuple (that list, [LP output
    (B, h, w, LP channels)]).  `taps`: optional dict filled with named intermediates (NCHW)."""
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.ascontiguousarray(x))
    x = x.float()
    conv = lambda c, t, ro=True: _conv_bn_act_torch(P, c, t, training, bn_stats, sim_bf16, ro)
    x = conv(g['stem'], x)
    if taps is not None:
        taps['stem'] = x
    routes = []
    n_st = len(g['stages'])
    for i, st in enumerate(g['stages']):
        x = conv(st['down'], x)
        for c1, c2 in st['res']:
            x = x + conv(c2, conv(c1, x), False)   # DarknetBasicBlockV3: no activation after add
            if sim_bf16:
                x = _bf16_round(x)
        if taps is not None:
            taps['stages.%d' % i] = x
        if i >= n_st - g['num_pyramid']:       # car/utils.py:73-74 (stem counts as stages[0] there)
            routes.append(x)
    outs = []
    lp_out = None
    for i, hd in enumerate(g['heads']):
        if 'lp' in g and i >= len(g['heads']) - 1:
            # CarLPNet.hybrid_forward, car_and_LP/YOLO.py:72-79: the LP branch reads the input of the finest block
            t = x
            for blk in g['lp']['blocks']:
                for c in blk['body']:
                    t = conv(c, t)
                t = conv(blk['tip'], t)                     # `_, LP_output = block(...)`: the tip feeds the next block
            lp_out = _conv_bn_act_torch(P, g['lp']['out'], t, sim_bf16=sim_bf16).permute(0, 2, 3, 1).contiguous()
        for c in hd['body']:
            x = conv(c, x)
        route = x
        tip = conv(hd['tip'], route)
        o = _conv_bn_act_torch(P, hd['out'], tip, sim_bf16=sim_bf16)
        outs.append(_yolo_output(o, hd['num_anchors'], g['per_anchor']))
        if i >= len(g['heads']) - 1:
            break
        x = conv(g['transitions'][i], route)
        x = torch.cat([_upsample2(x), routes[::-1][i + 1]], dim=1)
    if 'lp' in g:
        return outs[::-1], [lp_out]                          # car_and_LP/YOLO.py:95
    return outs[::-1]


def forward_torch_bf16sim(g, P, x):
    return forward_torch(g, P, x, sim_bf16=True)


def forward_torch_f16sim(g, P, x):
    """The same rounding points with IEEE half: the reference's use_fp16 path (net.cast('float16'), car/YOLO.py:98-100) as the
    HIP f16 path computes it -- fp16 weights and activations, fp32 accumulation, folded BN and LeakyReLU in fp32, one rounding
    per stored activation."""
    _SIM_DT[0] = torch.float16
    try:
        return forward_torch(g, P, x, sim_bf16=True)
    finally:
        _SIM_DT[0] = torch.bfloat16


# ----------------------------------------------------------------------------- numpy fp64
def _conv2d_np64(x, w, stride, pad):
    """Direct convolution: loop over the k*k taps, contract channels with einsum."""
    B, C, H, W = x.shape
    O, _, k, _ = w.shape
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    y = np.zeros((B, O, Ho, Wo), np.float64)
    for kh in range(k):
        for kw in range(k):
            patch = xp[:, :, kh:kh + stride * (Ho - 1) + 1:stride, kw:kw + stride * (Wo - 1) + 1:stride]
            y += np.einsum('bchw,oc->bohw', patch, w[:, :, kh, kw])
    return y


def _conv_bn_act_np64(P, c, x):
    w = np.asarray(P[c['name'] + '.weight'], np.float64)
    y = _conv2d_np64(x, w, c['stride'], c['pad'])
    if not c['bn']:
        return y + np.asarray(P[c['name'] + '.bias'], np.float64).reshape(1, -1, 1, 1)
    r = lambda n: np.asarray(P[c['name'] + '.' + n], np.float64).reshape(1, -1, 1, 1)
    y = (y - r('running_mean')) / np.sqrt(r('running_var') + BN_EPS) * r('gamma') + r('beta')
    return np.where(y > 0, y, LEAKY * y)


def forward_numpy64(g, P, x):
    x = np.asarray(x, np.float64)
    conv = lambda c, t: _conv_bn_act_np64(P, c, t)
    x = conv(g['stem'], x)
    routes = []
    n_st = len(g['stages'])
    for i, st in enumerate(g['stages']):
        x = conv(st['down'], x)
        for c1, c2 in st['res']:
            x = x + conv(c2, conv(c1, x))
        if i >= n_st - g['num_pyramid']:
            routes.append(x)
    outs = []
    lp_out = None
    for i, hd in enumerate(g['heads']):
        if 'lp' in g and i >= len(g['heads']) - 1:
            t = x
            for blk in g['lp']['blocks']:
                for c in blk['body']:
                    t = conv(c, t)
                t = conv(blk['tip'], t)
            lp_out = conv(g['lp']['out'], t).transpose(0, 2, 3, 1)
        for c in hd['body']:
            x = conv(c, x)
        route = x
        o = conv(hd['out'], conv(hd['tip'], route))
        B = o.shape[0]
        outs.append(o.transpose(0, 2, 3, 1).reshape(B, -1, hd['num_anchors'], g['per_anchor']))
        if i >= len(g['heads']) - 1:
            break
        x = conv(g['transitions'][i], route)
        x = np.concatenate([x.repeat(2, axis=-1).repeat(2, axis=-2), routes[::-1][i + 1]], axis=1)
    if 'lp' in g:
        return outs[::-1], [lp_out]
    return outs[::-1]
