"""Synthetic training targets: the label side of RenderCar.render (car/render_car.py:52-138)
and its GPU compositing step, and RenderCar.render itself for the PNG sprite set (host geometry with PIL as in the
reference, blend on the device).  The car sprites themselves (the PNG / PASCAL3D+ image sets, render_car.py:24,49-50)
are training data that is not part of the reference repository: RenderCar takes the directory they live in.

    label row = [cls, y, x, h, w, r, class distribution...]   (render_car.py:66-67,124-133)
    cls / distribution = get_label_dist(ele, azi)             (render_car.py:410-438)
    image = clip(bg / 255 * (1 - mask) + fg * mask, 0, 1)     (render_car.py:135-137)   -> yolo_composite (HIP)
"""
import math
import os
import random

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


# ---- RenderCar (car/render_car.py:29-138, 339-408): the host side of the synthetic-target generator -------------------
# The reference builds every training batch on the host with PIL (sprite -> random resize, rotate, blur -> paste at a random
# offset) and composites on the device; so does this class: PIL + numpy here, yolo_composite (HIP) for the blend.  The sprite
# sets themselves (blender renders named ...azi<1/100 deg>_ele<1/100 deg>.png under <root>/{train,valid}/<cad>/, and the
# PASCAL3D+ crops) are training data outside the reference repository: `root` points at a directory in that layout.  The
# sequence of np.random / random draws is the reference's, so a seeded run picks the same sprites, scales, angles and offsets.
PNG_MIN_SCALE, PNG_MAX_SCALE = 0.2, 1.0                         # render_car.py:20-21


class ColorAugmenter(object):
    """The colour part of mxnet.image.CreateAugmenter(data_shape, pca_noise=0.1, brightness=0.3, contrast=0.5,
    saturation=0.5, hue=1.0) as RenderCar uses it (render_car.py:45-47), restated from mxnet/image/image.py (mxnet is
    absent here -- [recalled]): ColorJitterAug (brightness, contrast, saturation in a random order) -> HueJitterAug ->
    LightingAug; the geometric augmenters of that list are no-ops for an image that already has the data shape.  Input and
    output: (H,W,3) float32 in 0..255."""
    COEF = np.array([[[0.299, 0.587, 0.114]]], np.float32)
    TYIQ = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.321], [0.211, -0.523, 0.311]])
    ITYIQ = np.array([[1.0, 0.956, 0.621], [1.0, -0.272, -0.647], [1.0, -1.107, 1.705]])
    EIGVAL = np.array([55.46, 4.794, 1.148])
    EIGVEC = np.array([[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140], [-0.5836, -0.6948, 0.4203]])

    def __init__(self, brightness=0.3, contrast=0.5, saturation=0.5, hue=1.0, pca_noise=0.1):
        self.b, self.c, self.s, self.h, self.pca = brightness, contrast, saturation, hue, pca_noise

    def _brightness(self, src):
        return src * np.float32(1.0 + random.uniform(-self.b, self.b))

    def _contrast(self, src):
        alpha = 1.0 + random.uniform(-self.c, self.c)
        gray = (3.0 * (1.0 - alpha) / src.size) * float((src * self.COEF).sum())
        return src * np.float32(alpha) + np.float32(gray)

    def _saturation(self, src):
        alpha = 1.0 + random.uniform(-self.s, self.s)
        gray = (src * self.COEF).sum(axis=2, keepdims=True) * np.float32(1.0 - alpha)
        return src * np.float32(alpha) + gray

    def __call__(self, src):
        src = np.asarray(src, np.float32)
        ts = [self._brightness, self._contrast, self._saturation]
        random.shuffle(ts)                                        # RandomOrderAug
        for t in ts:
            src = t(src)
        alpha = random.uniform(-self.h, self.h)                   # HueJitterAug
        u, w = math.cos(alpha * math.pi), math.sin(alpha * math.pi)
        bt = np.array([[1.0, 0.0, 0.0], [0.0, u, -w], [0.0, w, u]])
        src = src @ np.dot(np.dot(self.ITYIQ, bt), self.TYIQ).T.astype(np.float32)
        a = np.random.normal(0, self.pca, size=(3,))              # LightingAug
        return (src + np.dot(self.EIGVEC * a, self.EIGVAL).astype(np.float32)).astype(np.float32)


class RenderCar(object):
    """render_car.RenderCar(img_h, img_w, classes, ctx) for the PNG sprite set (`_render_png`; the PASCAL3D+ branch needs
    that data set's .mat annotations and is not built: pascal_rate must be 0)."""

    def __init__(self, img_h, img_w, classes, root, device='cuda:0', augment=True, R=30.0, G=0.3):
        self.h, self.w = int(img_h), int(img_w)
        self.classes = [list(c) for c in classes]
        self.num_cls = len(classes)
        self.device = device
        self.R, self.G = R, G                                     # PILImageEnhance(M=0, N=0, R=30.0, G=0.3, noise_var=0), :43-44
        self.augs = ColorAugmenter() if augment else None
        self.rawcar_dataset = {'train': [], 'valid': []}          # load_png_images, :187-219 (os.listdir order)
        for mode in self.rawcar_dataset:
            mdir = os.path.join(root, mode)
            if not os.path.isdir(mdir):
                continue
            for cad in os.listdir(mdir):
                for img in os.listdir(os.path.join(mdir, cad)):
                    self.rawcar_dataset[mode].append(os.path.join(mdir, cad, img))

    def _resize(self, pil_img, min_scale, max_scale, r1):
        """render_car.py:370-392."""
        from PIL import Image
        resize = np.random.uniform(low=min_scale, high=max_scale)
        resize_w = resize * pil_img.size[0]
        resize_h = resize * pil_img.size[1] * r1
        return resize, resize_w, resize_h, pil_img.resize((int(resize_w), int(resize_h)), Image.BILINEAR)

    def _enhance(self, img):
        """yolo_cv.PILImageEnhance.__call__ with M = N = 0, noise_var = 0 (yolo_cv.py:105-157): random_rotate, random_blur."""
        from PIL import Image, ImageFilter
        r = 0
        if self.R != 0:
            rd = np.random.uniform(low=-self.R, high=self.R)
            img = img.rotate(rd, Image.BILINEAR, expand=1)
            r = float(rd * np.pi) / 180
        if self.G != 0:
            img = img.filter(ImageFilter.GaussianBlur(radius=np.random.rand() * self.G))
        return img, r

    def _render_png(self, mode, r1=1.0):
        """render_car.py:339-368: one sprite -> (RGBA image, its bounding box after the rotation, r, class, distribution)."""
        from PIL import Image
        n = np.random.randint(len(self.rawcar_dataset[mode]))
        img_path = self.rawcar_dataset[mode][n]
        img = img_path.split('/')[-1]
        ele = float(img.split('ele')[1].split('.')[0]) * math.pi / 18000.
        azi = float(img.split('azi')[1].split('_')[0]) * math.pi / 18000.
        img_cls, label_distribution = get_label_dist(ele, azi, self.classes)
        pil_img = Image.open(img_path).convert('RGBA')
        _, _, _, pil_img = self._resize(pil_img, PNG_MIN_SCALE, PNG_MAX_SCALE, r1)
        pil_img, r = self._enhance(pil_img)
        box = pil_img.getbbox()
        if box is None:                                           # (a fully transparent sprite: the reference would fail here)
            box = (0, 0, pil_img.size[0], pil_img.size[1])
        return (pil_img,) + tuple(box) + (r, img_cls, label_distribution)

    def render_host(self, batch, mode, pascal_rate=0.0, render_rate=1.0):
        """The host half of render(): (fg (B,3,H,W) float32 0..1, mask (B,3,H,W) float32 0..1, labels (B,1,6+ncls))."""
        from PIL import Image
        if pascal_rate != 0.0:
            raise NotImplementedError('the PASCAL3D+ branch (_render_pascal) needs that data set: pascal_rate must be 0')
        fg = np.zeros((batch, 3, self.h, self.w), np.float32)
        mask = np.zeros((batch, 3, self.h, self.w), np.float32)
        labels = empty_labels(batch, self.num_cls)
        for i in range(batch):
            if np.random.rand() > render_rate:
                continue
            r1 = np.random.uniform(low=0.9, high=1.1)
            np.random.rand()                                      # (the draw compared with pascal_rate, :88)
            pil_img, l, t, r_, b, r, img_cls, dist = self._render_png(mode, r1)
            (xlo, xhi), (ylo, yhi) = paste_range(l, t, r_, b, self.h, self.w)
            paste_x = np.random.randint(low=xlo, high=xhi)
            paste_y = np.random.randint(low=ylo, high=yhi)
            tmp = Image.new('RGBA', (self.w, self.h))
            tmp.paste(pil_img, (paste_x, paste_y))
            rgb = np.asarray(Image.merge('RGB', tmp.split()[:3]), np.float32)          # pil_rgb_2_rgb_ndarray, yolo_gluon.py:303-313
            if self.augs is not None:
                rgb = self.augs(rgb)
            fg[i] = rgb.transpose(2, 0, 1) / np.float32(255.)
            m = np.asarray(tmp.split()[-1], np.float32) / np.float32(255.)              # pil_mask_2_rgb_ndarray, :298-300
            mask[i] = np.broadcast_to(m, (3, self.h, self.w))
            labels[i] = car_label(img_cls, l, t, r_, b, paste_x, paste_y, r, dist, self.h, self.w)
        return fg, mask, labels

    def render(self, bg, mode, pascal_rate=0.0, render_rate=1.0):
        """render_car.py:52-138: bg (B,3,H,W) float32 0..255 CUDA tensor -> (images 0..1 on the device, labels on the
        device); the blend clip(bg/255*(1-mask) + fg*mask, 0, 1) runs in yolo_composite."""
        import torch
        fg, mask, labels = self.render_host(len(bg), mode, pascal_rate, render_rate)
        dev = bg.device
        img = composite(bg, torch.from_numpy(fg).to(dev), torch.from_numpy(mask).to(dev))
        return img, torch.from_numpy(labels).to(dev)
