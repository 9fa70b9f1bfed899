"""Oracle, second witness (test infrastructure only): the decode / top-1 / IoU arithmetic as the reference states it a
SECOND time, in its insulator detector, restated here independently of oracle/detect.py (written from the insulator
files, structured the way they are: split the last axis, then op by op).  tests/test_witness.py requires the two
restatements -- and the HIP kernels -- to agree on the same inputs.

Restates
  YOLO.yxhw_to_ltrb   insulator/YOLO.py:306-321
  YOLO.predict        insulator/YOLO.py:323-341   (ONE image; arg-max over the flattened score; row [score,y,x,h,w,cls...])
  get_iou             insulator/utils.py:65-98    (mode 1 = cltrb, mode 2 = cyxhw; target_area = target[3]*target[4] :96)
  grid constants      insulator/YOLO.py:93-128    (steps -> s, y, x per cell; anchors h, w as fractions of the image)

PARITY UNPINNED (see oracle/__init__.py): mxnet's nd.sigmoid / nd.exp / nd.maximum are restated with numpy fp32.
"""
import numpy as np

F = np.float32


def _sig(v):
    return (F(1) / (F(1) + np.exp(-v.astype(F)))).astype(F)


def grid(size, steps, all_anchors):
    """insulator/YOLO.py:93-128 (the same construction car/YOLO.py:112-155 uses): per scale (fine -> coarse) and per
    cell in row-major order, s = stride in pixels, (y, x) = the cell's top-left corner in pixels, (h, w) = the anchor.
    Returns five (1, N, A, 1) float32 arrays."""
    S, Y, X, Hh, Ww = [], [], [], [], []
    for step, anchors in zip(steps, all_anchors):
        rows, cols, na = size[0] // step, size[1] // step, len(anchors)
        for r in range(rows):
            for c in range(cols):
                S.append([step] * na); Y.append([r * step] * na); X.append([c * step] * na)
                Hh.append([a[0] for a in anchors]); Ww.append([a[1] for a in anchors])
    sh = lambda v: np.asarray(v, F).reshape(1, -1, len(all_anchors[0]), 1)
    return sh(S), sh(Y), sh(X), sh(Hh), sh(Ww)


def yxhw_to_ltrb(yxhw, size, consts):
    """insulator/YOLO.py:306-321."""
    s, y0, x0, h0, w0 = consts
    ty, tx, th, tw = [yxhw[..., k:k + 1].astype(F) for k in range(4)]        # .split(num_outputs=4, axis=-1)
    by = (_sig(ty) * s + y0) / F(size[0])
    bx = (_sig(tx) * s + x0) / F(size[1])
    bh = np.exp(th) * h0
    bw = np.exp(tw) * w0
    bh2 = bh / F(2)
    bw2 = bw / F(2)
    left = bx - bw2
    right = bx + bw2
    top = by - bh2
    bottom = by + bh2
    return np.concatenate([left, top, right, bottom], axis=-1).astype(F)


def predict(score_logits, yxhw, rest, size, consts):
    """insulator/YOLO.py:323-341: score (1,N,A,1), yxhw (1,N,A,4), rest (1,N,A,K) of ONE image ->
    [sigmoid(score), y, x, h, w, rest...] of the box with the highest score (first index among ties)."""
    c_score = _sig(score_logits)
    c_box = yxhw_to_ltrb(yxhw, size, consts)
    cout = np.concatenate([c_score, c_box, rest.astype(F)], axis=-1)
    cout = cout.reshape(-1, 5 + rest.shape[-1])
    best = int(np.argmax(c_score.reshape(-1)))
    row = cout[best].copy()
    y = (row[2] + row[4]) / F(2)
    x = (row[1] + row[3]) / F(2)
    h = row[4] - row[2]
    w = row[3] - row[1]
    row[1:5] = [y, x, h, w]
    return row, best


def get_iou(predict_ltrb, target, mode=1):
    """insulator/utils.py:65-98."""
    p = np.asarray(predict_ltrb, F)
    target = np.asarray(target, F)
    l, t, r, b = [p[..., k:k + 1] for k in range(4)]
    if mode == 1:
        l2, t2, r2, b2 = target[1], target[2], target[3], target[4]
    elif mode == 2:
        l2 = target[2] - target[4] / F(2)
        t2 = target[1] - target[3] / F(2)
        r2 = target[2] + target[4] / F(2)
        b2 = target[1] + target[3] / F(2)
    else:
        raise ValueError('mode should be int 1 or 2')
    i_left, i_top = np.maximum(l2, l), np.maximum(t2, t)
    i_right, i_bottom = np.minimum(r2, r), np.minimum(b2, b)
    iw = np.maximum(i_right - i_left, F(0))
    ih = np.maximum(i_bottom - i_top, F(0))
    inters = iw * ih
    predict_area = (r - l) * (b - t)
    target_area = target[3] * target[4]
    with np.errstate(divide='ignore', invalid='ignore'):
        return (inters / (predict_area + target_area - inters)).astype(F)
