"""Oracle: anchor grids, decode, top-1, IoU, NMS, LPD plumbing (test infrastructure only).

numpy fp32 restatement (the reference computes these in fp32 NDArray ops) of
  _init_step/_init_area/_init_syxhw   car/YOLO.py:112-155
  _get_default_ltrb                   car/YOLO.py:209-240
  merge_and_slice                     car/YOLO.py:841-849
  _yxhw_to_ltrb                       car/YOLO.py:552-566
  predict                             car/YOLO.py:568-597
  get_iou                             yolo_modules/yolo_gluon.py:127-168
  cv_img_2_ndarray                    yolo_modules/yolo_gluon.py:335-357
  predict_LP                          licence_plate/LP_detection.py:147-162
  nms                                 NOT IN THE REFERENCE; defined here per
                                      SURVEY App. A.8 (mxnet contrib.box_nms semantics as used by
                                      gluoncv YOLOv3): the only reference-derived invariant is
                                      kept[0] == argmax(sigmoid(obj)) in objectness mode.

PARITY UNPINNED (see oracle/__init__.py).
"""
import math
import numpy as np

f32 = np.float32


def sigmoid(x):
    x = np.asarray(x, f32)
    return (f32(1) / (f32(1) + np.exp(-x))).astype(f32)


def init_steps(layers, all_anchors):
    """car/YOLO.py:112-116."""
    nd_, np_ = len(layers), len(all_anchors)
    start = nd_ - np_ + 1
    return [2 ** (start + i) for i in range(np_)]


def init_area(size, steps):
    """car/YOLO.py:118-121."""
    return [int(size[0] * size[1] / step ** 2) for step in steps]


def init_syxhw(size, steps, all_anchors):
    """car/YOLO.py:123-155.  Returns s,y,x,h,w each (1, sum(area), A, 1) float32.
    Scale order fine->coarse, cells row-major, anchor innermost."""
    area = init_area(size, steps)
    n = len(all_anchors[0])
    tot = sum(area)
    s = np.zeros((1, tot, n, 1), f32); y = s.copy(); x = s.copy(); h = s.copy(); w = s.copy()
    a0 = 0
    for i, anchors in enumerate(all_anchors):
        a, step = area[i], steps[i]
        xn, yn = int(size[1] / step), int(size[0] / step)
        ys = np.repeat(np.arange(0, size[0], step, dtype=f32), n * xn)           # nd.arange(repeat=n*x_num)
        xs = np.tile(np.repeat(np.arange(0, size[1], step, dtype=f32), n), yn)
        hw = np.tile(np.asarray(anchors, f32), (a, 1))
        s[0, a0:a0 + a] = f32(step)
        y[0, a0:a0 + a] = ys.reshape(a, n, 1)
        x[0, a0:a0 + a] = xs.reshape(a, n, 1)
        h[0, a0:a0 + a] = hw[:, 0].reshape(a, n, 1)
        w[0, a0:a0 + a] = hw[:, 1].reshape(a, n, 1)
        a0 += a
    return s, y, x, h, w


def get_default_ltrb(size, steps, all_anchors):
    """car/YOLO.py:209-240: anchor boxes centred on cell centres, normalised ltrb (sum(area), A, 4)."""
    out = []
    for i, anchors in enumerate(all_anchors):
        anchors = np.asarray(anchors, f32)
        n = len(anchors)
        step = float(steps[i])
        yn, xn = int(size[0] / step), int(size[1] / step)
        a = yn * xn
        hh, ww = anchors[:, 0], anchors[:, 1]
        yc = (np.arange(yn, dtype=f32) * f32(step / size[0]) + f32(step / size[0] / 2.))   # nd.arange(start, 1, step)
        y = np.repeat(yc, n * xn)
        h = np.tile(hh, a)
        top = (y - f32(0.5) * h).reshape(a, n, 1)
        bot = (y + f32(0.5) * h).reshape(a, n, 1)
        xc = (np.arange(xn, dtype=f32) * f32(step / size[1]) + f32(step / size[1] / 2.))
        x = np.repeat(xc, n)
        w = np.tile(ww, xn)
        left = np.tile(x - f32(0.5) * w, yn).reshape(a, n, 1)
        right = np.tile(x + f32(0.5) * w, yn).reshape(a, n, 1)
        out.append(np.concatenate([left, top, right, bot], axis=-1))
    return np.concatenate(out, axis=0).astype(f32)


def merge_and_slice(all_output, points):
    """car/YOLO.py:841-849."""
    out = np.concatenate([np.asarray(o) for o in all_output], axis=1)
    res, i = [], 0
    for pt in points:
        res.append(out[..., i:pt])
        i = pt
    return res


def yxhw_to_ltrb(yxhw, size, syxhw):
    """car/YOLO.py:552-566.  yxhw (B,N,A,4) raw [ty,tx,th,tw] -> normalised [l,t,r,b]."""
    s, y, x, h, w = syxhw
    ty, tx, th, tw = [yxhw[..., k:k + 1].astype(f32) for k in range(4)]
    by = (sigmoid(ty) * s + y) / f32(size[0])
    bx = (sigmoid(tx) * s + x) / f32(size[1])
    bh = np.exp(th) * h
    bw = np.exp(tw) * w
    bh2, bw2 = bh / f32(2), bw / f32(2)
    return np.concatenate([bx - bw2, by - bh2, bx + bw2, by + bh2], axis=-1).astype(f32)


def decode_all(batch_out, slice_point, size, syxhw):
    """Rows [sigmoid(obj), l,t,r,b, rot_raw, cls_logits...] for every box: (B, N*A, 6+ncls).
    car/YOLO.py:571-579."""
    sl = merge_and_slice(batch_out, slice_point)
    # car specs: slice_point [1,3,5,6,30] -> score, yx, hw, rot, cls
    score = sigmoid(sl[0])
    box = yxhw_to_ltrb(np.concatenate([sl[1], sl[2]], axis=-1), size, syxhw)
    rows = np.concatenate([score, box, sl[3], sl[4]], axis=-1).astype(f32)
    B = rows.shape[0]
    return rows.reshape(B, -1, rows.shape[-1])


def predict(batch_out, slice_point, size, syxhw):
    """car/YOLO.py:568-597 -> (B, 6+ncls) rows [score, y, x, h, w, rot, cls...]; also returns
    the arg-max flat box index per image (lowest index among ties, mxnet argmax)."""
    rows = decode_all(batch_out, slice_point, size, syxhw)
    B = rows.shape[0]
    pred = np.zeros((B, rows.shape[-1]), f32)
    idx = np.zeros(B, np.int64)
    for i in range(B):
        k = int(np.argmax(rows[i, :, 0]))
        p = rows[i, k].copy()
        y = (p[2] + p[4]) / f32(2); x = (p[1] + p[3]) / f32(2)
        h = p[4] - p[2]; w = p[3] - p[1]
        p[1:5] = [y, x, h, w]
        pred[i] = p
        idx[i] = k
    return pred, idx


def get_iou(predict_ltrb, target, mode=2):
    """yolo_gluon.py:127-168.  predict (...,4) ltrb ; target (5,) [c,y,x,h,w] (mode 2) or
    [c,l,t,r,b] (mode 1, incl. the reference's target_area = target[3]*target[4] quirk :166)."""
    p = np.asarray(predict_ltrb, f32)
    t = np.asarray(target, f32)
    l, tt, r, b = [p[..., k:k + 1] for k in range(4)]
    if mode == 1:
        l2, t2, r2, b2 = t[1], t[2], t[3], t[4]
    else:
        l2 = t[2] - t[4] / f32(2); t2 = t[1] - t[3] / f32(2)
        r2 = t[2] + t[4] / f32(2); b2 = t[1] + t[3] / f32(2)
    iw = np.maximum(np.minimum(r2, r) - np.maximum(l2, l), f32(0))
    ih = np.maximum(np.minimum(b2, b) - np.maximum(t2, tt), f32(0))
    inter = iw * ih
    pa = (r - l) * (b - tt)
    ta = t[3] * t[4]
    with np.errstate(divide='ignore', invalid='ignore'):
        return (inter / (pa + ta - inter)).astype(f32)


def box_iou_ltrb(a, b):
    """IoU between two ltrb boxes, max(0,.) intersections, no +1, fp32."""
    iw = max(f32(0), min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(f32(0), min(a[3], b[3]) - max(a[1], b[1]))
    inter = f32(iw) * f32(ih)
    ua = f32(f32(a[2] - a[0]) * f32(a[3] - a[1])) + f32(f32(b[2] - b[0]) * f32(b[3] - b[1])) - inter
    return f32(inter / ua) if ua > 0 else f32(0)


def softmax(x, axis=-1):
    x = np.asarray(x, f32)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True)).astype(f32)


def nms(rows, mode='class', valid_thresh=0.01, iou_thresh=0.45, topk=400, post_nms=100, scores=None):
    """Per-image greedy NMS over decoded rows (N*A, 6+ncls) (from decode_all).
    mode 'class': candidates = every (box, class) pair, score = sigmoid(obj)*softmax(cls)_c,
                  candidate id = box*ncls + c; suppression only within the same class.
    mode 'obj'  : candidates = boxes, score = sigmoid(obj), class-agnostic.
    Order: stable sort by score descending (ties -> lower candidate id first); drop score <
    valid_thresh; keep the first topk; greedy: j suppressed by an earlier kept i (same class)
    iff IoU(i,j) > iou_thresh (strict); at most post_nms kept.  Returns (ids, scores).
    `scores`: optional precomputed candidate scores (flat, candidate-id order) used instead of
    recomputing them, so index parity can be checked on bit-identical inputs."""
    rows = np.asarray(rows, f32)
    nbox = rows.shape[0]
    if mode == 'obj':
        if scores is None:
            scores = rows[:, 0].copy()
        cls_of = np.zeros(nbox, np.int64)
        box_of = np.arange(nbox)
    else:
        ncls = rows.shape[1] - 6
        if scores is None:
            prob = softmax(rows[:, 6:], axis=-1)
            scores = (rows[:, 0:1] * prob).astype(f32).reshape(-1)
        cls_of = np.tile(np.arange(ncls), nbox)
        box_of = np.repeat(np.arange(nbox), ncls)
    scores = np.asarray(scores, f32).reshape(-1)
    cand = np.nonzero(scores >= f32(valid_thresh))[0]
    order = cand[np.argsort(-scores[cand], kind='stable')][:topk]
    kept = []
    for j in order:
        bj = rows[box_of[j], 1:5]
        ok = True
        for i in kept:
            if cls_of[i] != cls_of[j]:
                continue
            if box_iou_ltrb(rows[box_of[i], 1:5], bj) > f32(iou_thresh):
                ok = False
                break
        if ok:
            kept.append(int(j))
            if len(kept) >= post_nms:
                break
    kept = np.asarray(kept, np.int64)
    return kept, scores[kept] if len(kept) else np.zeros(0, f32)


def cv_img_2_ndarray(image):
    """yolo_gluon.py:335-357 without the optional resize: (H,W,3) uint8 -> (1,3,H,W) float32 /255.
    Channel order is left as delivered (the reference does not swap BGR->RGB)."""
    a = np.asarray(image).astype(f32)
    return (a.transpose(2, 0, 1)[None] / f32(255.)).astype(f32)


def predict_LP(batch_out, r_max):
    """licence_plate/LP_detection.py:147-162.  batch_out (1,10,h,w) -> (10,) pose row."""
    out = np.asarray(batch_out, f32).transpose(0, 2, 3, 1)[0]
    best = int(np.argmax(out[:, :, 0].reshape(-1)))
    pred = out.reshape(-1, out.shape[-1])[best].copy()
    pred[0] = sigmoid(pred[0])
    pred[1:4] *= f32(1000)
    for i in range(3):
        p = (sigmoid(pred[i + 4]) - f32(0.5)) * f32(2) * f32(r_max[i])
        pred[i + 4] = p * f32(math.pi) / f32(180.)
    return pred, best


def predict_LP_batch(LP_batch_out, LP_slice_point, r_max):
    """car_and_LP/YOLO.py:133-169 (predict_LP + LP_pose_activation).  LP_batch_out: [ (B,h,w,C) ] as CarLPNet
    returns it.  merge_and_slice at LP_slice_point -> score / xy / z / r (the LP class slice is dropped); per image
    the cell with the highest sigmoid(score) (first among ties) -> [sigmoid(score), x*1000, y*1000, z*1000, three angles
    (sigmoid - 0.5) * 2 * r_max * pi / 180].  Returns (B,7) float32 and the chosen cell indices."""
    out = np.concatenate([np.asarray(o, f32) for o in LP_batch_out], axis=1)
    B, C = out.shape[0], out.shape[-1]
    n_used = LP_slice_point[3]                       # score 1 + xy 2 + z 1 + r 3 = 7
    out = out.reshape(B, -1, C)
    preds, best = np.zeros((B, n_used), f32), np.zeros(B, np.int64)
    for b in range(B):
        score = sigmoid(out[b, :, 0])
        k = int(np.argmax(score))
        # sigmoid is monotone but not injective in float32: the reference takes the arg-max of the SIGMOID
        best[b] = k
        p = out[b, k, :n_used].copy()
        p[0] = score[k]
        p[1:4] *= f32(1000)
        for i in range(3):
            v = (sigmoid(p[4 + i]) - f32(0.5)) * f32(2) * f32(r_max[i])
            p[4 + i] = v * f32(math.pi) / f32(180.)
        preds[b] = p
    return preds, best
