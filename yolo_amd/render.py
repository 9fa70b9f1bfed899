"""Synthetic training targets: the label side of RenderCar.render (car/render_car.py:52-138)
and its GPU compositing step.  The car sprites themselves (the PNG / PASCAL3D+ image sets, render_car.py:24,49-50)
are training data that is not part of the reference repository, so the sprite is an input here.

    label row = [cls, y, x, h, w, r, class distribution...]   (render_car.py:66-67,124-133)
    cls / distribution = get_label_dist(ele, azi)             (render_car.py:410-438)
    image = clip(bg / 255 * (1 - mask) + fg * mask, 0, 1)     (render_car.py:135-137)   -> yolo_composite (HIP)
"""
import math

import numpy as np


def get_label_dist(ele, azi, classes, sigma=0.1):
    """render_car.py:410-438: great-circle angle between (ele, azi) [rad] and every class direction
    (`classes` rows = [azimuth deg, elevation deg], spec.yaml `classes`), Gaussian in that angle, normalised.
    Returns (arg-min class, float32 distribution)."""
    cl = np.asarray(classes, np.float64)
    azi_l, ele_l = np.deg2rad(cl[:, 0]), np.deg2rad(cl[:, 1])
    ang = np.arccos(np.clip(math.sin(ele) * np.sin(ele_l) + math.cos(ele) * np.cos(ele_l) * np.cos(azi - azi_l), -1, 1))
    g = np.exp(-(ang.astype(np.float32)) ** 2 / np.float32(sigma))
    return int(np.argmin(ang)), (g / g.sum()).astype(np.float32)


def paste_range(r_box_l, r_box_t, r_box_r, r_box_b, img_h, img_w):
    """render_car.py:101-108: the integer ranges [low, high) the paste offsets are drawn from, so that at least
    70 % of the rotated sprite box stays inside the image."""
    w, h = r_box_r - r_box_l, r_box_b - r_box_t
    return ((int(-r_box_l - 0.3 * w), int(img_w - r_box_l - 0.7 * w)),
            (int(-r_box_t - 0.3 * h), int(img_h - r_box_t - 0.7 * h)))


def car_label(img_cls, r_box_l, r_box_t, r_box_r, r_box_b, paste_x, paste_y, r, label_distribution, img_h, img_w):
    """render_car.py:110-133: (1, 6+ncls) label [cls, y, x, h, w (fractions of the image), r, distribution]."""
    box_y = (r_box_b + r_box_t) / 2. + paste_y
    box_x = (r_box_r + r_box_l) / 2. + paste_x
    box_h, box_w = float(r_box_b - r_box_t), float(r_box_r - r_box_l)
    head = np.asarray([img_cls, box_y / img_h, box_x / img_w, box_h / img_h, box_w / img_w, r], np.float32)
    return np.concatenate([head, np.asarray(label_distribution, np.float32).reshape(-1)])[None]


def empty_labels(batch, num_class):
    """render_car.py:80: rows of -1 = 'no object' (skipped by _loss_mask, car/YOLO.py:468)."""
    return -np.ones((batch, 1, 6 + num_class), np.float32)


def composite(bg, fg, mask):
    """render_car.py:135-137 on device: bg (B,3,H,W) float32 0..255, fg / mask 0..1 CUDA tensors -> images 0..1."""
    import torch
    from . import lib as L
    bg, fg, mask = bg.contiguous(), fg.contiguous(), mask.contiguous()
    if not (bg.shape == fg.shape == mask.shape) or bg.dtype != torch.float32:
        raise ValueError('bg, fg and mask must be float32 tensors of one shape')
    out = torch.empty_like(bg)
    L.check(L.load().yolo_composite(L.ptr(bg), L.ptr(fg), L.ptr(mask), L.ptr(out), bg.numel(), L.stream_ptr()), 'composite')
    return out
