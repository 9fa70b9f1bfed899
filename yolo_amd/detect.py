"""Detection post-processing: host-side mirror of the decode / top-1 / IoU part of car/YOLO.py
and yolo_modules/yolo_gluon.py, backed by the HIP kernels of csrc/detect.hip.

  Detector.predict(outs)       <- YOLO.predict            car/YOLO.py:568-597 (owns the D2H copy)
  Detector.decode(outs)        <- _yxhw_to_ltrb + concat  car/YOLO.py:552-579
  Detector.nms(rows, ...)      <- new: the reference has no NMS
  get_iou(pred, target, mode)  <- yolo_gluon.get_iou      yolo_gluon.py:127-168
  cv_img_2_ndarray(img)        <- yolo_gluon.py:335-357
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L


def make_grid(all_anchors, size, steps):
    """Anchor-grid descriptor (car/YOLO.py:112-155): scales fine->coarse, cells row-major, anchor innermost."""
    g = L.GridDesc()
    g.nscale, g.A = len(all_anchors), len(all_anchors[0])
    if g.nscale > 4 or g.A > 8:
        raise ValueError('at most 4 scales x 8 anchors')
    g.img_h, g.img_w = int(size[0]), int(size[1])
    nbox = 0
    for i, anchors in enumerate(all_anchors):
        if len(anchors) != g.A:
            raise ValueError('every scale must have the same number of anchors')
        g.gh[i], g.gw[i], g.step[i] = int(size[0] / steps[i]), int(size[1] / steps[i]), int(steps[i])
        nbox += g.gh[i] * g.gw[i] * g.A
        for a, (h, w) in enumerate(anchors):
            g.anchors_hw[(i * g.A + a) * 2 + 0] = float(h)
            g.anchors_hw[(i * g.A + a) * 2 + 1] = float(w)
    return g, nbox


class Detector(object):
    def __init__(self, spec, size, steps, device='cuda:0'):
        self.spec, self.size = spec, (int(size[0]), int(size[1]))
        self.grid, self.nbox = make_grid(spec['all_anchors'], size, steps)
        sp = spec['slice_point']
        # channel order obj, ty, tx, th, tw, rot, cls... (car/v1/spec.yaml:6 slice_point [1,3,5,6,30])
        if list(sp[:4]) != [1, 3, 5, 6]:
            raise ValueError('decode expects slice_point [1,3,5,6,C]')
        self.C = sp[-1]
        self.device = L.resolve_device(device)
        self._lib = L.load()
        self._nms_ws = {}
        self._host = {}

    def _merged(self, outs):
        """Accepts the list of per-scale outputs (fine->coarse) or an already merged (B,N,A,C) tensor."""
        L.require_current_device(self.device, 'this Detector')
        if isinstance(outs, (list, tuple)):
            base = outs[0]._base
            if (base is not None and all(o._base is base for o in outs) and base.dim() == 3
                    and base.is_contiguous() and base.shape[1] * self.grid.A == self.nbox):
                # CarNet returns views into one merged (B, sum HW, A*C) buffer: no concat copy needed
                m = base.view(base.shape[0], base.shape[1], self.grid.A, self.C)
            else:
                m = torch.cat(list(outs), dim=1)           # merge_and_slice's concat, car/YOLO.py:842
        else:
            m = outs
        m = m.contiguous()
        if m.dtype != torch.float32 or m.shape[1] * m.shape[2] != self.nbox or m.shape[3] != self.C:
            raise ValueError('expected (B,%d,A,%d) float32 logits' % (self.nbox // self.grid.A, self.C))
        return m

    def decode(self, outs):
        """-> rows (B, N*A, C) float32 on device: [sigmoid(obj), l, t, r, b, rot, cls...]."""
        m = self._merged(outs)
        B = m.shape[0]
        rows = torch.empty((B, self.nbox, self.C), dtype=torch.float32, device=m.device)
        L.check(self._lib.yolo_decode(L.ptr(m), L.ptr(rows), B, self.C, C.byref(self.grid), L.stream_ptr()), 'decode')
        return rows

    def decode_scores(self, outs, mode='class'):
        """decode() and nms_scores() in one pass over the logits -> (rows, scores); bit-identical to the two calls."""
        m = self._merged(outs)
        B = m.shape[0]
        md = 1 if mode == 'class' else 0
        rows = torch.empty((B, self.nbox, self.C), dtype=torch.float32, device=m.device)
        scores = torch.empty((B, self.nbox * ((self.C - 6) if md else 1)), dtype=torch.float32, device=m.device)
        L.check(self._lib.yolo_decode_scores(L.ptr(m), L.ptr(rows), L.ptr(scores), B, self.C, C.byref(self.grid), md,
                                             L.stream_ptr()), 'decode_scores')
        return rows, scores

    def decode_nms(self, outs, mode='class', valid_thresh=0.01, iou_thresh=0.45, topk=400, post_nms=100):
        """decode_scores() + nms() as one library call (yolo_decode_nms): the decode pass also takes the selection's first
        histogram.  -> (rows, scores, kept ids, kept scores, kept count); identical to the two calls."""
        m = self._merged(outs)
        B = m.shape[0]
        md = 1 if mode == 'class' else 0
        rows = torch.empty((B, self.nbox, self.C), dtype=torch.float32, device=m.device)
        scores = torch.empty((B, self.nbox * ((self.C - 6) if md else 1)), dtype=torch.float32, device=m.device)
        ws = self._nms_ws.get(B)
        if ws is None:
            ws = self._nms_ws[B] = torch.empty(self._lib.yolo_nms_select_workspace_bytes(B), dtype=torch.uint8, device=m.device)
        kept = torch.empty((B, post_nms), dtype=torch.int32, device=m.device)
        ks = torch.empty((B, post_nms), dtype=torch.float32, device=m.device)
        cnt = torch.empty((B,), dtype=torch.int32, device=m.device)
        L.check(self._lib.yolo_decode_nms(L.ptr(m), L.ptr(rows), L.ptr(scores), B, self.C, C.byref(self.grid), md, valid_thresh,
                                          iou_thresh, topk, post_nms, L.ptr(kept), L.ptr(ks), L.ptr(cnt), L.ptr(ws),
                                          L.stream_ptr()), 'decode_nms')
        return rows, scores, kept, ks, cnt

    def predict_device(self, outs):
        m = self._merged(outs)
        B = m.shape[0]
        pred = torch.empty((B, self.C), dtype=torch.float32, device=m.device)
        idx = torch.empty((B,), dtype=torch.int32, device=m.device)
        L.check(self._lib.yolo_predict_top1(L.ptr(m), L.ptr(pred), L.ptr(idx), B, self.C, C.byref(self.grid),
                                            L.stream_ptr()), 'predict_top1')
        return pred, idx

    def predict(self, outs):
        """YOLO.predict: np.float32 (B, 6+ncls) rows [score, y, x, h, w, rot, cls...] (D2H here)."""
        pred, _ = self.predict_device(outs)
        return pred.cpu().numpy()

    def predict_async(self, outs, slot=0):
        """predict() without the host stall: the (B, 6+ncls) rows are copied into a pinned host buffer by an asynchronous
        copy ordered behind the kernels on the current stream.  -> (rows, event): `rows` is a np.float32 view of the pinned
        buffer of `slot`, valid once `event.synchronize()` returned and until the next call with the same slot.  Two slots
        let the copy of frame i overlap the launches of frame i + 1 (bench.py's step)."""
        pred, _ = self.predict_device(outs)
        key = (slot, tuple(pred.shape))
        host = self._host.get(key)
        if host is None:
            host = self._host[key] = torch.empty(pred.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(pred, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host.numpy(), ev

    def nms_scores(self, rows, mode='class'):
        B, nbox, Cc = rows.shape
        md = 1 if mode == 'class' else 0
        ncand = nbox * ((Cc - 6) if md else 1)
        scores = torch.empty((B, ncand), dtype=torch.float32, device=rows.device)
        L.check(self._lib.yolo_nms_scores(L.ptr(rows), L.ptr(scores), B, nbox, Cc, md, L.stream_ptr()), 'nms_scores')
        return scores

    def nms(self, rows, mode='class', valid_thresh=0.01, iou_thresh=0.45, topk=400, post_nms=100, scores=None, fast=True):
        """Greedy NMS.  Returns (kept ids (B,post_nms) int32 padded with -1,
        kept scores, kept count).  Candidate id = box*ncls + class in 'class' mode, box in 'obj' mode."""
        rows = rows.contiguous()
        B, nbox, Cc = rows.shape
        cpb = (Cc - 6) if mode == 'class' else 1
        if scores is None:
            scores = self.nms_scores(rows, mode)
        ws = self._nms_ws.get(B) if fast else None
        if fast and ws is None:
            ws = self._nms_ws[B] = torch.empty(self._lib.yolo_nms_select_workspace_bytes(B), dtype=torch.uint8, device=rows.device)
        kept = torch.empty((B, post_nms), dtype=torch.int32, device=rows.device)
        ks = torch.empty((B, post_nms), dtype=torch.float32, device=rows.device)
        cnt = torch.empty((B,), dtype=torch.int32, device=rows.device)
        L.check(self._lib.yolo_nms_from_scores(L.ptr(rows), L.ptr(scores), B, nbox, Cc, cpb, valid_thresh, iou_thresh,
                                               topk, post_nms, L.ptr(kept), L.ptr(ks), L.ptr(cnt), L.ptr(ws), L.stream_ptr()),
                'nms')
        return kept, ks, cnt


def get_iou(predict, target, mode=1):
    """yolo_gluon.get_iou (yolo_gluon.py:127-168): predict (...,4) ltrb CUDA float32 vs ONE target (5,).
    mode 1 (the reference's default): target = [c, l, t, r, b], with the reference's target_area = target[3] *
    target[4] (:166) kept; mode 2 (what the hot path passes, car/YOLO.py:403,525): target = [c, y, x, h, w]."""
    if mode not in (1, 2):
        raise ValueError('mode should be int 1 or 2')              # (the reference only prints this, then fails on l2)
    p = predict.contiguous().view(-1, 4)
    t = target.to(p.device, torch.float32).contiguous()
    out = torch.empty((p.shape[0],), dtype=torch.float32, device=p.device)
    lib = L.load()
    fn = lib.yolo_iou_ltrb_vs_cltrb if mode == 1 else lib.yolo_iou_ltrb_vs_yxhw
    L.check(fn(L.ptr(p), L.ptr(t), L.ptr(out), p.shape[0], L.stream_ptr()), 'iou')
    return out.view(tuple(predict.shape[:-1]) + (1,))


def cv_img_2_ndarray(image, device='cuda:0'):
    """(H,W,3) uint8 ndarray -> (1,3,H,W) float32 /255 on device; channel order untouched."""
    img = torch.from_numpy(np.ascontiguousarray(image)).to(device)
    H, W, Cc = img.shape
    out = torch.empty((1, Cc, H, W), dtype=torch.float32, device=img.device)
    L.check(L.load().yolo_image_u8_to_nchw(L.ptr(img), L.ptr(out), 1, H, W, Cc, L.stream_ptr()), 'image_u8_to_nchw')
    return out


def default_ltrb(all_anchors, size, steps):
    """_get_default_ltrb (car/YOLO.py:209-240): anchor boxes centred on the cell centres, normalised
    [l,t,r,b], shape (sum(area), A, 4) float32, scales fine->coarse.  Host-side constant (the reference
    builds it once in _init_train); fp32 op order: centre = index*pitch + pitch/2, edge = centre -+ 0.5*size."""
    f32 = np.float32
    out = []
    for i, anchors in enumerate(all_anchors):
        an = np.asarray(anchors, f32)
        n = len(an)
        step = float(steps[i])
        yn, xn = int(size[0] / step), int(size[1] / step)
        a = yn * xn
        yc = np.arange(yn, dtype=f32) * f32(step / size[0]) + f32(step / size[0] / 2.)
        xc = np.arange(xn, dtype=f32) * f32(step / size[1]) + f32(step / size[1] / 2.)
        y = np.repeat(yc, n * xn)
        h = np.tile(an[:, 0], a)
        x = np.repeat(xc, n)
        w = np.tile(an[:, 1], xn)
        top, bot = (y - f32(0.5) * h).reshape(a, n, 1), (y + f32(0.5) * h).reshape(a, n, 1)
        left = np.tile(x - f32(0.5) * w, yn).reshape(a, n, 1)
        right = np.tile(x + f32(0.5) * w, yn).reshape(a, n, 1)
        out.append(np.concatenate([left, top, right, bot], axis=-1))
    return np.concatenate(out, axis=0).astype(f32)


def predict_LP(batch_out, r_max):
    """LicencePlateDetectioin.predict_LP (licence_plate/LP_detection.py:147-162): (1,C,h,w) float32 CUDA
    tensor -> np.float32 (C,) pose row of the best cell."""
    o = batch_out.contiguous()
    _, Cc, h, w = o.shape
    pred = torch.empty((Cc,), dtype=torch.float32, device=o.device)
    idx = torch.empty((1,), dtype=torch.int32, device=o.device)
    L.check(L.load().yolo_predict_lp(L.ptr(o), L.ptr(pred), L.ptr(idx), Cc, h, w, float(r_max[0]), float(r_max[1]),
                                     float(r_max[2]), L.stream_ptr()), 'predict_lp')
    return pred.cpu().numpy()


def predict_LP_batch(LP_batch_out, LP_slice_point, r_max):
    """CarLPNet's predict_LP (car_and_LP/YOLO.py:133-169): [ (B,h,w,C) float32 CUDA ] -> np.float32 (B,7) rows
    [sigmoid(score), x, y, z (x1000), r1, r2, r3 (rad)] of the best cell of every image."""
    if list(LP_slice_point[:4]) != [1, 3, 4, 7]:
        raise ValueError('predict_LP expects LP_slice_point [1,3,4,7,C]')
    o = LP_batch_out[0] if isinstance(LP_batch_out, (list, tuple)) else LP_batch_out
    o = o.contiguous()
    B, h, w, Cc = o.shape
    pred = torch.empty((B, 7), dtype=torch.float32, device=o.device)
    idx = torch.empty((B,), dtype=torch.int32, device=o.device)
    L.check(L.load().yolo_predict_lp_nhwc(L.ptr(o), L.ptr(pred), L.ptr(idx), B, h * w, Cc, float(r_max[0]),
                                          float(r_max[1]), float(r_max[2]), L.stream_ptr()), 'predict_lp_nhwc')
    return pred.cpu().numpy()

