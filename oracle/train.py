"""Oracle: target assignment, losses, optimiser step, batch sharding (test infrastructure only).

Restates
  _find_best          car/YOLO.py:401-448
  _loss_mask          car/YOLO.py:450-480
  _score_weight       car/YOLO.py:482-489
  _get_loss           car/YOLO.py:491-498  (gluon LogisticLoss(binary), HuberLoss(rho=1),
                      SoftmaxCrossEntropyLoss(sparse_label=False) restated from their published
                      formulas -- mxnet is absent from /root/reference)
  _train_batch        car/YOLO.py:350-399  (sum(losses).backward(); trainer.step(batch_size))
  mxnet Adam          SURVEY App. A.6 (epsilon outside the bias correction; rescale_grad = 1/batch)
  split_render_data   yolo_modules/yolo_gluon.py:100-124
  get_label_dist      car/render_car.py:410-438 (synthetic label generator for config 3)

PARITY UNPINNED (see oracle/__init__.py).
"""
import math
import numpy as np
import torch

from . import detect
from .forward import forward_torch

f32 = np.float32


def inv_sigmoid(x):
    """yolo_gluon.py:365."""
    return -np.log(f32(1) / x - f32(1))


def find_best(L, anchors_ltrb, all_anchors, size, steps, area):
    """car/YOLO.py:401-448.  L: (6+C,) label [cls,y,x,h,w,r,...].  Returns (pixel, anchor, [ty,tx,th,tw])."""
    L = np.asarray(L, f32)
    A = len(all_anchors[0])
    ious = detect.get_iou(anchors_ltrb, L, mode=2)
    best = int(np.argmax(ious.reshape(-1)))
    px, anc = best // A, best % A
    if px >= sum(area):
        px = sum(area) - 1
    ltrb = anchors_ltrb[px, anc]
    a0 = 0
    for i, a in enumerate(area):
        a0 += a
        if px < a0:
            layer = i
            break
    step = f32(steps[layer])
    sty = (L[1] - (ltrb[3] + ltrb[1]) / f32(2)) * f32(size[0]) / step + f32(0.5)
    sty = np.clip(sty, f32(0.0001), f32(0.9999))
    ty = inv_sigmoid(f32(sty))
    stx = (L[2] - (ltrb[2] + ltrb[0]) / f32(2)) * f32(size[1]) / step + f32(0.5)
    stx = np.clip(stx, f32(0.0001), f32(0.9999))
    tx = inv_sigmoid(f32(stx))
    th = np.log(L[3] / f32(all_anchors[layer][anc][0]))
    tw = np.log(L[4] / f32(all_anchors[layer][anc][1]))
    return px, anc, np.asarray([ty, tx, th, tw], f32)


def loss_mask(labels, anchors_ltrb, all_anchors, size, steps, area, num_class):
    """car/YOLO.py:450-480.  labels (B, nobj, 6+C).  Returns ([score,yx,hw,rot,cls], mask)."""
    labels = np.asarray(labels, f32)
    bs, a, n = labels.shape[0], sum(area), len(all_anchors[0])
    mask = np.zeros((bs, a, n, 1), f32)
    score = np.zeros((bs, a, n, 1), f32)
    yx = np.zeros((bs, a, n, 2), f32)
    hw = np.zeros((bs, a, n, 2), f32)
    rot = np.zeros((bs, a, n, 1), f32)
    cls = np.zeros((bs, a, n, num_class), f32)
    for b in range(bs):
        for L in labels[b]:
            if L[0] < 0:
                continue
            px, anc, box = find_best(L, anchors_ltrb, all_anchors, size, steps, area)
            mask[b, px, anc] = 1.0
            score[b, px, anc] = 1.0
            yx[b, px, anc] = box[:2]
            hw[b, px, anc] = box[2:]
            rot[b, px, anc] = L[5]
            cls[b, px, anc] = L[6:]
    return [score, yx, hw, rot, cls], mask


def score_weight(mask, positive_weight=1.0, negative_weight=0.1):
    """car/YOLO.py:482-489."""
    return np.where(mask > 0, f32(positive_weight), f32(negative_weight)).astype(f32)


# gluon losses -- restated; all reduce with mean over every non-batch axis
def _mean_nb(t):
    return t.reshape(t.shape[0], -1).mean(dim=1)

def logistic_loss(pred, label, w):
    loss = torch.relu(pred) - pred * label + torch.log1p(torch.exp(-torch.abs(pred)))   # binary format
    return _mean_nb(loss * w)

def huber_loss(pred, label, w, rho=1.0):
    d = torch.abs(label - pred)
    loss = torch.where(d > rho, d - 0.5 * rho, (0.5 / rho) * d * d)
    return _mean_nb(loss * w)

def softmax_ce_loss(pred, label, w):
    loss = -(torch.log_softmax(pred, dim=-1) * label).sum(dim=-1, keepdim=True)
    return _mean_nb(loss * w)


DEFAULT_SCALE = {'score': 0.1, 'box_yx': 0.01, 'box_hw': 10.0, 'rotate': 0.0, 'class': 0.3}  # car/v1/spec.yaml:31-35


def get_loss(x, y, s_weight, mask, scale=DEFAULT_SCALE, car_rotate=False):
    """car/YOLO.py:491-498.  x: 5 torch tensors (sliced net output); y: 5 targets; returns 5 x (B,)."""
    T = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    y = [T(v) for v in y]
    s_weight, mask = T(s_weight), T(mask)
    rot = scale['rotate'] if car_rotate else 0
    s = logistic_loss(x[0], y[0], s_weight * scale['score'])
    yx = huber_loss(x[1], y[1], mask * scale['box_yx'])
    hw = huber_loss(x[2], y[2], mask * scale['box_hw'])
    r = huber_loss(x[3], y[3], mask * rot)
    c = softmax_ce_loss(x[4], y[4], mask * scale['class'])
    return s, yx, hw, r, c


def loss_and_grad_wrt_output(merged_out, labels, spec, size, scale=DEFAULT_SCALE,
                             positive_weight=1.0, negative_weight=0.1):
    """Loss vectors and d(sum of all losses)/d(net output) for a merged (B,N,A,C) fp32 output."""
    steps = detect.init_steps(spec['layers'], spec['all_anchors'])
    area = detect.init_area(size, steps)
    anchors_ltrb = detect.get_default_ltrb(size, steps, spec['all_anchors'])
    sp = spec['slice_point']
    ncls = sp[-1] - sp[-2]
    y, mask = loss_mask(labels, anchors_ltrb, spec['all_anchors'], size, steps, area, ncls)
    sw = score_weight(mask, positive_weight, negative_weight)
    out = torch.from_numpy(np.ascontiguousarray(merged_out)).float().requires_grad_(True)
    xs, i = [], 0
    for pt in sp:
        xs.append(out[..., i:pt]); i = pt
    losses = get_loss(xs, y, sw, mask, scale)
    total = sum(l.sum() for l in losses)      # sum(losses).backward() on (B,) vectors = sum over batch
    total.backward()
    return [l.detach().numpy() for l in losses], out.grad.numpy(), (y, mask, sw)


def adam_step(w, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, rescale=1.0):
    """mxnet Adam.  t is the 1-based update count.  In-place on numpy fp32 arrays."""
    g = (g * f32(rescale)).astype(f32)
    # (1 - beta) is taken in fp32 FROM the fp32 beta, as MXNet's adam_update kernel does (`scalar<DType>(1.f - param.beta1)`
    # [recalled: src/operator/optimizer_op-inl.h]): 1.f - 0.999f = 0.00099998713, not float(0.001) -- a 1.3e-5 relative
    # difference in v that the K-step trajectory test (tests/test_gpu_trajectory.py) is tight enough to see
    m[...] = f32(beta1) * m + (f32(1) - f32(beta1)) * g
    v[...] = f32(beta2) * v + (f32(1) - f32(beta2)) * g * g
    lr_t = f32(lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    w[...] = w - lr_t * m / (np.sqrt(v) + f32(eps))
    return w, m, v


def split_render_data(batch, n_ctx):
    """yolo_gluon.py:100-124."""
    bs = len(batch)
    return [batch[int(i * bs / n_ctx):int((i + 1) * bs / n_ctx)] for i in range(n_ctx)]


def get_label_dist(ele, azi, classes, sigma=0.1):
    """car/render_car.py:410-438.  classes: list of [azi_deg, ele_deg]."""
    cl = np.asarray(classes, np.float64)
    azi_l, ele_l = np.deg2rad(cl[:, 0]), np.deg2rad(cl[:, 1])
    ang = np.arccos(np.clip(math.sin(ele) * np.sin(ele_l) + math.cos(ele) * np.cos(ele_l) * np.cos(azi - azi_l), -1, 1))
    g = np.exp(-(ang.astype(f32)) ** 2 / f32(sigma))
    return int(np.argmin(ang)), (g / g.sum()).astype(f32)


def synthetic_labels(batch, seed=3, render_rate=0.5, num_class=24):
    """Config-3 synthetic targets: (B,1,6+C); rows of -1 mean 'no object'."""
    rng = np.random.default_rng(seed)
    classes = [[15.0 * i, 0.0] for i in range(num_class)]          # car/v1/spec.yaml:13-19
    lab = -np.ones((batch, 1, 6 + num_class), f32)
    for b in range(batch):
        if rng.random() < render_rate:
            continue
        azi = rng.uniform(0, 2 * math.pi)
        c, dist = get_label_dist(0.0, azi, classes)
        y, x = rng.uniform(.15, .85, 2)
        h, w = rng.uniform(.2, .9, 2)
        r = rng.uniform(-30, 30) * math.pi / 180
        lab[b, 0, :6] = [c, y, x, h, w, r]
        lab[b, 0, 6:] = dist
    return lab


def train_step_reference(g, P, x, labels, spec, size, scale=DEFAULT_SCALE, sim_bf16=False):
    """One forward (train-mode BN) + loss + backward on the torch-CPU graph; returns losses and
    gradients w.r.t. every trainable parameter (dict name -> ndarray)."""
    Pt = {}
    for k, v in P.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).clone()
        if k.endswith(('.weight', '.gamma', '.beta', '.bias')):
            t.requires_grad_(True)
        Pt[k] = t
    # sim_bf16: forward with the bf16 training path's rounding points (straight-through in the backward)
    outs = forward_torch(g, Pt, x, training=True, sim_bf16=sim_bf16)
    merged = torch.cat(outs, dim=1)
    steps = detect.init_steps(spec['layers'], spec['all_anchors'])
    area = detect.init_area(size, steps)
    anchors_ltrb = detect.get_default_ltrb(size, steps, spec['all_anchors'])
    sp = spec['slice_point']
    y, mask = loss_mask(labels, anchors_ltrb, spec['all_anchors'], size, steps, area, sp[-1] - sp[-2])
    sw = score_weight(mask)
    xs, i = [], 0
    for pt in sp:
        xs.append(merged[..., i:pt]); i = pt
    losses = get_loss(xs, y, sw, mask, scale)
    sum(l.sum() for l in losses).backward()
    grads = {k: t.grad.numpy() for k, t in Pt.items() if t.requires_grad and t.grad is not None}
    return [l.detach().numpy() for l in losses], grads, merged.detach().numpy()


# ---- CarLPNet: the licence-plate losses (licence_plate/LP_detection.py:258-360, car_and_LP/YOLO.py:262-300) ---------
LP_DEFAULT_SCALE = {'LP_score': 0.1, 'LP_xy': 10.0, 'LP_z': 1.0, 'LP_r': 0.1, 'LP_class': 0.0}   # car_and_LP/v1/spec.yaml


def find_best_LP(L, size, step, r_max):
    """LP_detection.py:258-280.  L: (10,) LP label [flag, X, Y, Z (mm), r1, r2, r3 (rad), x_px, y_px, type].
    Returns ((h_feature, w_feature), [tX, tY, tZ, tr1, tr2, tr3])."""
    L = np.asarray(L, f32)
    h_max, w_max = size[0] // step - 1, size[1] // step - 1
    h_f = int(np.clip(int(L[8] / f32(step)), 0, h_max))
    w_f = int(np.clip(int(L[7] / f32(step)), 0, w_max))
    t = [L[1] / f32(1000.), L[2] / f32(1000.), L[3] / f32(1000.)]
    for i in range(3):
        rm = f32(r_max[i] * math.pi / 180.)
        t.append(inv_sigmoid(L[4 + i] / rm / f32(2.) + f32(0.5)))
    return (h_f, w_f), np.asarray(t, f32)


def loss_mask_LP(labels, size, step, r_max, num_class):
    """LP_detection.py:282-312.  labels (B, nobj, 10).  Returns ([score, xy, z, r, cls], mask), each (B, h, w, k)."""
    labels = np.asarray(labels, f32)
    bs, h_, w_ = labels.shape[0], size[0] // step, size[1] // step
    score = np.zeros((bs, h_, w_, 1), f32); mask = np.zeros((bs, h_, w_, 1), f32)
    xy = np.zeros((bs, h_, w_, 2), f32); z = np.zeros((bs, h_, w_, 1), f32)
    r = np.zeros((bs, h_, w_, 3), f32); cls = np.zeros((bs, h_, w_, num_class), f32)
    for b in range(bs):
        for L in labels[b]:
            if L[0] < 0:
                continue
            (hf, wf), p = find_best_LP(L, size, step, r_max)
            score[b, hf, wf] = 1.0; mask[b, hf, wf] = 1.0
            xy[b, hf, wf] = p[:2]; z[b, hf, wf] = p[2]; r[b, hf, wf] = p[3:]
            cls[b, hf, wf, int(L[-1])] = 1
    return [score, xy, z, r, cls], mask


def get_loss_LP(x, y, s_weight, mask, scale=LP_DEFAULT_SCALE):
    """LP_detection.py:354-360."""
    T = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    y = [T(v) for v in y]
    s_weight, mask = T(s_weight), T(mask)
    s = logistic_loss(x[0], y[0], s_weight * scale['LP_score'])
    xy = huber_loss(x[1], y[1], mask * scale['LP_xy'])
    z = huber_loss(x[2], y[2], mask * scale['LP_z'])
    r = huber_loss(x[3], y[3], mask * scale['LP_r'])
    c = softmax_ce_loss(x[4], y[4], mask * scale['LP_class'])
    return s, xy, z, r, c


def lp_loss_and_grad_wrt_output(lp_out, lp_labels, size, step, r_max, slice_point, scale=LP_DEFAULT_SCALE,
                                positive_weight=1.0, negative_weight=0.1):
    """Loss vectors and d(sum of LP losses)/d(LP output) for an (B,h,w,C) fp32 LP-branch output."""
    ncls = slice_point[-1] - slice_point[-2]
    y, mask = loss_mask_LP(lp_labels, size, step, r_max, ncls)
    sw = score_weight(mask, positive_weight, negative_weight)
    out = torch.from_numpy(np.ascontiguousarray(lp_out)).float().requires_grad_(True)
    xs, i = [], 0
    for pt in slice_point:
        xs.append(out[..., i:pt]); i = pt
    losses = get_loss_LP(xs, y, sw, mask, scale)
    sum(l.sum() for l in losses).backward()
    return [l.detach().numpy() for l in losses], out.grad.numpy(), (y, mask)


def synthetic_lp_labels(batch, size, seed=4, add_rate=0.5, r_max=(45, 60, 45), num_class=3):
    """LP targets in LPGenerator.add's layout (licence_plate_render/__init__.py:134-166): (B,1,10)
    [1, X, Y, Z (mm), r1, r2, r3 (rad), x_px, y_px, type]; rows of -1 = no plate."""
    rng = np.random.default_rng(seed)
    lab = -np.ones((batch, 1, 10), f32)
    for b in range(batch):
        if rng.random() > add_rate:
            continue
        r = [rng.uniform(-0.9, 0.9) * rm * math.pi / 180. for rm in r_max]
        lab[b, 0] = [1, rng.uniform(-2000, 2000), rng.uniform(-1000, 1000), rng.uniform(2000, 9000), r[0], r[1], r[2],
                     rng.uniform(0, size[1]), rng.uniform(0, size[0]), rng.integers(0, num_class)]
    return lab


def train_step_reference_lp(g, P, x, labels, lp_labels, spec, size, scale=DEFAULT_SCALE, lp_scale=LP_DEFAULT_SCALE,
                            sim_bf16=False):
    """CarLPNet's _train_batch (car_and_LP/YOLO.py:262-300): the five car losses + the five LP losses, one backward."""
    Pt = {}
    for k, v in P.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).clone()
        if k.endswith(('.weight', '.gamma', '.beta', '.bias')):
            t.requires_grad_(True)
        Pt[k] = t
    outs, lp = forward_torch(g, Pt, x, training=True, sim_bf16=sim_bf16)
    merged = torch.cat(outs, dim=1)
    steps = detect.init_steps(spec['layers'], spec['all_anchors'])
    area = detect.init_area(size, steps)
    anchors_ltrb = detect.get_default_ltrb(size, steps, spec['all_anchors'])
    sp = spec['slice_point']
    y, mask = loss_mask(labels, anchors_ltrb, spec['all_anchors'], size, steps, area, sp[-1] - sp[-2])
    xs, i = [], 0
    for pt in sp:
        xs.append(merged[..., i:pt]); i = pt
    losses = list(get_loss(xs, y, score_weight(mask), mask, scale))
    lsp = spec['LP_slice_point']
    ly, lmask = loss_mask_LP(lp_labels, size, steps[0], spec.get('LP_r_max', [45, 60, 45]), lsp[-1] - lsp[-2])
    lxs, i = [], 0
    for pt in lsp:
        lxs.append(lp[0][..., i:pt]); i = pt
    losses += list(get_loss_LP(lxs, ly, score_weight(lmask, spec.get('LP_positive_weight', 1.0),
                                                     spec.get('LP_negative_weight', 0.1)), lmask, lp_scale))
    sum(l.sum() for l in losses).backward()
    grads = {k: t.grad.numpy() for k, t in Pt.items() if t.requires_grad and t.grad is not None}
    return [l.detach().numpy() for l in losses], grads, merged.detach().numpy(), lp[0].detach().numpy()



# ---- one-hop backward references from SAVED forward values (tests/test_gpu_configs.py) ---------------------------------
# A randomly initialised net in train mode is chaotic (a 1e-5 forward difference flips LeakyReLU' signs), so comparing a
# whole backward pass end to end needs loose bars.  These helpers restate the backward of ONE layer of _train_batch's
# autograd graph (car/YOLO.py:381-392; gluoncv _conv2d = Convolution -> BatchNorm(train) -> LeakyReLU(0.1), SURVEY
# App. A.1/A.3) given the values the layer under test itself saved: errors cannot accumulate across layers and every
# gradient of the step can be held to a tight bar.
def bn_act_backward(yraw, gamma, beta, dz, eps=1e-5, slope=0.1, flip=None):
    """Backward of train-mode BatchNorm (biased batch variance) + LeakyReLU at the saved raw convolution output `yraw`
    (B,C,H,W), evaluated in float64.  Returns (z, dy, dgamma, dbeta, a) with a = the BatchNorm output (the LeakyReLU
    input); a residual branch adds to z and receives dz unchanged.

    flip: optional bool mask of elements whose LeakyReLU derivative is taken on the OTHER side of the kink.  An element
    whose `a` is within rounding of zero lands on either side depending on the operation order of the implementation
    under test; the caller passes the (few, checked to be near-kink) elements where that happened so that dy, dgamma
    and dbeta are compared for the same branch decisions."""
    y = torch.as_tensor(yraw).double()
    g = torch.as_tensor(gamma).double().view(1, -1, 1, 1)
    b = torch.as_tensor(beta).double().view(1, -1, 1, 1)
    dz = torch.as_tensor(dz).double()
    mean = y.mean(dim=(0, 2, 3), keepdim=True)
    var = y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    invstd = 1.0 / torch.sqrt(var + eps)
    xhat = (y - mean) * invstd
    a = xhat * g + b
    z = torch.where(a > 0, a, a * slope)
    d = torch.where(a > 0, torch.ones_like(a), torch.full_like(a, slope))
    if flip is not None:
        d = torch.where(torch.as_tensor(flip), (1.0 + slope) - d, d)
    da = dz * d
    dbeta = da.sum(dim=(0, 2, 3))
    dgamma = (da * xhat).sum(dim=(0, 2, 3))
    n = y.shape[0] * y.shape[2] * y.shape[3]
    dy = g * invstd * (da - dbeta.view(1, -1, 1, 1) / n - xhat * dgamma.view(1, -1, 1, 1) / n)
    return z, dy, dgamma, dbeta, a


def conv_backward(x, w, dy, stride):
    """Data and weight gradient of Convolution(no bias, pad = k // 2) at the saved input `x` for the given dy."""
    x, w, dy = torch.as_tensor(x), torch.as_tensor(w), torch.as_tensor(dy)
    pad = w.shape[2] // 2
    dx = torch.nn.grad.conv2d_input(x.shape, w, dy, stride=stride, padding=pad)
    dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=stride, padding=pad)
    return dx, dw
