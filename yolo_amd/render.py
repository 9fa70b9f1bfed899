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


def composite(bg, fg, mask, unit_bg=False):
    """render_car.py:135-137 on device: bg (B,3,H,W) float32 0..255, fg / mask 0..1 CUDA tensors -> images 0..1.
    unit_bg: bg is already 0..1 (LPGenerator.add's blend, licence_plate_render/__init__.py:163)."""
    import torch
    from . import lib as L
    bg, fg, mask = bg.contiguous(), fg.contiguous(), mask.contiguous()
    if not (bg.shape == fg.shape == mask.shape) or bg.dtype != torch.float32:
        raise ValueError('bg, fg and mask must be float32 tensors of one shape')
    out = torch.empty_like(bg)
    fn = L.load().yolo_composite_unit if unit_bg else L.load().yolo_composite
    L.check(fn(L.ptr(bg), L.ptr(fg), L.ptr(mask), L.ptr(out), bg.numel(), L.stream_ptr()), 'composite')
    return out


# ---- RenderCar (car/render_car.py:29-138, 339-408): the host side of the synthetic-target generator -------------------
# The reference builds every training batch on the host with PIL (sprite -> random resize, rotate, blur -> paste at a random
# offset) and composites on the device; so does this class: PIL + numpy here, yolo_composite (HIP) for the blend.  The sprite
# sets themselves (blender renders named ...azi<1/100 deg>_ele<1/100 deg>.png under <root>/{train,valid}/<cad>/, and the
# PASCAL3D+ crops) are training data outside the reference repository: `root` points at a directory in that layout.  The
# sequence of np.random / random draws is the reference's, so a seeded run picks the same sprites, scales, angles and offsets.
PNG_MIN_SCALE, PNG_MAX_SCALE = 0.2, 1.0                         # render_car.py:20-21
PASCAL_MIN_SCALE, PASCAL_MAX_SCALE = 0.2, 0.9                   # render_car.py:23-24


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


def pascal3d_view(mat):
    """get_pascal3d_azi_ele (render_car.py:440-458) on one loaded PASCAL3D+ annotation (the dict scipy.io.loadmat returns):
    -> (elevation rad, azimuth rad, [l, t, r, b]) of the image's single car, or None when the image holds several
    (the reference skips those).  The record's fields are addressed by POSITION, as the reference does: record[1] =
    objects, object[1] = bbox, object[3] = viewpoint, viewpoint[2] / [3] = azimuth / elevation in degrees."""
    objects = mat['record'][0][0][1][0]
    if len(objects) != 1:
        return None
    view = objects[0][3][0][0]
    return (float(np.ravel(view[3])[0]) * math.pi / 180., float(np.ravel(view[2])[0]) * math.pi / 180., [int(v) for v in objects[0][1][0]])


class RenderCar(object):
    """render_car.RenderCar(img_h, img_w, classes, ctx): the PNG sprite set (`_render_png`) and, when `pascal_root` is
    given, the PASCAL3D+ crops (`_render_pascal`; <pascal_root>/car_imagenet_label/*.mat, car_imagenet_{train,valid}/,
    pre-loaded as the reference does with pre_load=True, render_car.py:221-260)."""

    def __init__(self, img_h, img_w, classes, root, device='cuda:0', augment=True, R=30.0, G=0.3, pascal_root=None):
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
        self.pascal_dataset = {'train': [], 'valid': []}          # load_pascal_images, :221-260
        if pascal_root is not None:
            from PIL import Image
            import scipy.io as sio
            ldir = os.path.join(pascal_root, 'car_imagenet_label')
            anno = {f: sio.loadmat(os.path.join(ldir, f)) for f in os.listdir(ldir)}
            for mode in self.pascal_dataset:
                idir = os.path.join(pascal_root, 'car_imagenet_' + mode)
                for img in os.listdir(idir):
                    view = pascal3d_view(anno[img.split('.')[0] + '.mat'])
                    if view is None:
                        continue
                    cls, dist = get_label_dist(view[0], view[1], self.classes)
                    self.pascal_dataset[mode].append((Image.open(os.path.join(idir, img)).convert('RGBA'), view[2], cls, dist))

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
        # the box of pixels that are non-zero in ANY band, as the PIL of the reference's day computed it (Pillow >= 10 looks
        # at the alpha band only by default; bilinear rotation leaves colour in a rim of fully transparent pixels, so the
        # two differ by a pixel or two -- found by the oracle comparison, tests/test_render.py)
        try:
            box = pil_img.getbbox(alpha_only=False)
        except TypeError:
            box = pil_img.getbbox()
        if box is None:                                           # (a fully transparent sprite: the reference would fail here)
            box = (0, 0, pil_img.size[0], pil_img.size[1])
        return (pil_img,) + tuple(box) + (r, img_cls, label_distribution)

    def _render_pascal(self, mode, r1=1.0):
        """render_car.py:262-337: a PASCAL3D+ crop scaled so that its annotated car box spans 20-90 % of the image; the box
        itself -- not the alpha channel: the crops are opaque photographs -- is carried through the resize and the
        (zero-degree, see below) rotation to give the label box."""
        from PIL import Image, ImageFilter
        data = self.pascal_dataset[mode]
        sprite, box, img_cls, dist = data[np.random.randint(len(data))]
        box = np.asarray(box, np.float64)                         # l, t, r, b
        span_w, span_h = box[2] - box[0], (box[3] - box[1]) * r1
        hi = min(PASCAL_MAX_SCALE * self.w / span_w, PASCAL_MAX_SCALE * self.h / span_h)
        lo = max(PASCAL_MIN_SCALE * self.w / span_w, PASCAL_MIN_SCALE * self.h / span_h)
        scale, new_w, new_h, sprite = self._resize(sprite, lo, hi, r1)
        # pil_image_enhance(pil_img, R=0) (:306): the enhancer was built with R = 30, so its rotation step still runs, with
        # the call's R = 0: ONE uniform(-0, 0) draw and a rotation by 0 degrees; then the blur
        deg = np.random.uniform(low=-0.0, high=0.0)
        sprite = sprite.rotate(deg, Image.BILINEAR, expand=1)
        r = float(deg * np.pi) / 180
        if self.G != 0:
            sprite = sprite.filter(ImageFilter.GaussianBlur(radius=np.random.rand() * self.G))
        # box corners relative to the image centre -> rotated by r -> relative to the corner of the expanded canvas
        cx = box[[0, 2]] * scale - 0.5 * new_w
        cy = box[[1, 3]] * scale * r1 - 0.5 * new_h
        gx, gy = np.meshgrid(cx, cy, indexing='ij')
        rx, ry = gx * math.cos(r) - gy * math.sin(r), gy * math.cos(r) + gx * math.sin(r)
        half_w = 0.5 * (abs(new_h * math.sin(r)) + abs(new_w * math.cos(r)))
        half_h = 0.5 * (abs(new_h * math.cos(r)) + abs(new_w * math.sin(r)))
        return (sprite, rx.min() + half_w, ry.min() + half_h, rx.max() + half_w, ry.max() + half_h, r, img_cls, dist)

    def render_host(self, batch, mode, pascal_rate=0.0, render_rate=1.0):
        """The host half of render(): (fg (B,3,H,W) float32 0..1, mask (B,3,H,W) float32 0..1, labels (B,1,6+ncls))."""
        from PIL import Image
        if pascal_rate != 0.0 and not self.pascal_dataset[mode]:
            raise ValueError('pascal_rate > 0 needs the PASCAL3D+ crops: RenderCar(..., pascal_root=...)')
        fg = np.zeros((batch, 3, self.h, self.w), np.float32)
        mask = np.zeros((batch, 3, self.h, self.w), np.float32)
        labels = empty_labels(batch, self.num_cls)
        for i in range(batch):
            if np.random.rand() > render_rate:
                continue
            r1 = np.random.uniform(low=0.9, high=1.1)
            if np.random.rand() < pascal_rate:                    # (:88; the draw is made whatever the rate)
                pil_img, l, t, r_, b, r, img_cls, dist = self._render_pascal(mode, r1)
            else:
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


# ---- LPGenerator.add (yolo_modules/licence_plate_render/__init__.py:21-166, 273-371): licence plates for CarLPNet ---------
LP_CORNERS = np.float32([[380, 160], [0, 160], [0, 0], [380, 0]])       # (:118: the plate image's corners, as projected)
LP_GLYPH_X = (7, 56, 106, 158, 175, 225, 274, 324)                      # (:28: glyph columns of the 'ABC-1234' plate)


def homography(src, dst):
    """cv2.getPerspectiveTransform(src, dst) (cv2 is absent here): the projective map through four point pairs, as the
    3x3 matrix normalised to M[2, 2] = 1 -- the eight unknowns of  u = (a x + b y + c) / (g x + h y + 1),
    v = (d x + e y + f) / (g x + h y + 1)  from the eight linear equations the pairs give."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    x, y, u, v = src[:, 0], src[:, 1], dst[:, 0], dst[:, 1]
    one, zero = np.ones(4), np.zeros(4)
    A = np.concatenate([np.stack([x, y, one, zero, zero, zero, -x * u, -y * u], axis=1),
                        np.stack([zero, zero, zero, x, y, one, -x * v, -y * v], axis=1)])
    return np.append(np.linalg.solve(A, np.concatenate([u, v])), 1.0).reshape(3, 3)


class PlateCamera(object):
    """ProjectRectangle6D (:273-371): the pinhole camera the plates are projected through.  `camera`: the calibration
    the reference reads from its camera yaml -- image_width, image_height, projection_matrix.data (row-major 3x4)."""
    HALF_W, HALF_H = 199.5, 84.0                                  # (the reference's constants, mm)

    def __init__(self, camera):
        self.w, self.h = int(camera['image_width']), int(camera['image_height'])
        P = camera['projection_matrix']['data']
        self.fx, self.fy, self.cx, self.cy = float(P[0]), float(P[5]), float(P[2]), float(P[6])

    def corners(self, pose):
        """Pixel positions of the plate's corners (bottom-right, bottom-left, top-left, top-right as the reference orders
        them) for pose [X, Y, Z mm, r1, r2, r3 rad]: K (R3 R2 R1 P + T) in closed form (:337-363)."""
        X, Y, Z, r1, r2, r3 = [float(v) for v in pose]
        Rx = np.array([[1, 0, 0], [0, math.cos(r1), -math.sin(r1)], [0, math.sin(r1), math.cos(r1)]])
        Ry = np.array([[math.cos(r2), 0, math.sin(r2)], [0, 1, 0], [-math.sin(r2), 0, math.cos(r2)]])
        Rz = np.array([[math.cos(r3), -math.sin(r3), 0], [math.sin(r3), math.cos(r3), 0], [0, 0, 1]])
        P = np.array([[self.HALF_W, -self.HALF_W, -self.HALF_W, self.HALF_W], [self.HALF_H, self.HALF_H, -self.HALF_H, -self.HALF_H],
                      [0.0, 0.0, 0.0, 0.0]])
        cam = Rz @ Ry @ Rx @ P + np.array([[X], [Y], [Z]])
        K = np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]])
        pix = K @ cam
        return (pix[:2] / pix[2]).T.astype(np.float32)

    def centre(self, X, Y, Z, out_h, out_w):
        """(:126-130) the plate centre in pixels of the (out_h, out_w) training image."""
        return ((X * self.fx / Z + self.cx) * out_w / float(self.w), (Y * self.fy / Z + self.cy) * out_h / float(self.h))


class LPGenerator(object):
    """LPGenerator(img_h, img_w) (:21-56) for `add` (:134-166): draws an 'ABC-1234' plate from glyph images, projects it
    with a random 6-D pose through the camera, blurs / noises it and pastes it onto a batch of 0..1 images on the device.
    `fonts_dir` holds the glyphs 0.png .. 33.png (digits, then letters) and the dot 34.png as the reference's
    licence_plate_render/fonts does -- data that stays with the reference; `camera`: see PlateCamera."""

    def __init__(self, img_h, img_w, fonts_dir, camera, augment=True):
        from PIL import Image
        self.h, self.w = int(img_h), int(img_w)
        self.camera = PlateCamera(camera)
        self.glyph = [Image.open(os.path.join(fonts_dir, '%d.png' % k)).resize((45, 90), Image.BILINEAR) for k in range(34)]
        self.dot = Image.open(os.path.join(fonts_dir, '34.png')).resize((10, 70), Image.BILINEAR)
        # (:52-55: augs2 = CreateAugmenter(pca_noise=0.1, brightness=0.7, contrast=0.7, saturation=0.7, hue=1.0))
        self.augs = ColorAugmenter(brightness=0.7, contrast=0.7, saturation=0.7, hue=1.0, pca_noise=0.1) if augment else None

    def draw_LP(self):
        """(:58-77) -> (RGBA plate 380 x 160 on white, type 0, [[glyph id, left, right (fractions of the width)] x 7])."""
        from PIL import Image
        plate = Image.new('RGBA', (380, 160), (255, 255, 255))
        letters = np.random.randint(10, 34, size=3)
        ids = list(letters)
        for k, g in enumerate(letters):
            plate.paste(self.glyph[g], (LP_GLYPH_X[k], 35))
        plate.paste(self.dot, (LP_GLYPH_X[3], 45))
        digits = np.random.randint(0, 9, size=4)
        for k, g in enumerate(digits):
            g = 9 if g == 4 else g                                # (no digit four on a plate)
            ids.append(g)
            plate.paste(self.glyph[g], (LP_GLYPH_X[k + 4], 35))
        cols = LP_GLYPH_X[:3] + LP_GLYPH_X[4:]
        return plate, 0, [[int(g), c / 380., (c + 45) / 380.] for g, c in zip(ids, cols)]

    def random_projection_LP_6D(self, plate, out_size, r_max):
        """(:98-132) -> (mask (3,H,W), image (3,H,W) float32 0..1, label [1, X, Y, Z, r1, r2, r3, x_px, y_px])."""
        from PIL import Image, ImageFilter
        Z = np.random.uniform(low=1500., high=5000.)
        X = (Z * 9 / 30.) * np.random.uniform(low=-1, high=1)
        Y = (Z * 7 / 30.) * np.random.uniform(low=-1, high=1)
        rot = [np.random.uniform(low=-1, high=1) * r_max[k] * math.pi / 180. for k in range(3)]
        M = homography(self.camera.corners([X, Y, Z] + rot), LP_CORNERS)
        plate = plate.transform((self.camera.w, self.camera.h), Image.PERSPECTIVE, tuple(M.reshape(-1)[:8]), Image.BILINEAR)
        plate = plate.resize((out_size[1], out_size[0]), Image.BILINEAR)
        # pil_image_enhance(LP, G=1.0, noise_var=5.0) of PILImageEnhance(M=0, N=0, R=0, G=1.0, noise_var=10.): blur, noise
        plate = plate.filter(ImageFilter.GaussianBlur(radius=np.random.rand() * 1.0))
        px = np.array(plate)
        plate = Image.fromarray(np.uint8(np.clip(px + np.random.normal(0., 5.0, px.shape), 0, 255)))
        bands = plate.split()
        rgb = np.asarray(Image.merge('RGB', bands[:3]), np.float32)
        if self.augs is not None:
            rgb = self.augs(rgb)
        alpha = np.asarray(bands[-1], np.float32) / np.float32(255.)
        x, y = self.camera.centre(X, Y, Z, out_size[0], out_size[1])
        label = np.asarray([1, X, Y, Z, rot[0], rot[1], rot[2], x, y], np.float32)
        return np.broadcast_to(alpha, (3,) + alpha.shape), rgb.transpose(2, 0, 1) / np.float32(255.), label

    def add_host(self, batch, h, w, r_max, add_rate=1.0):
        """The host half of add(): (fg, mask (B,3,h,w) float32, labels (B,1,10) [1, X, Y, Z, r1, r2, r3, x, y, type]; -1: none)."""
        fg = np.zeros((batch, 3, h, w), np.float32)
        mask = np.zeros((batch, 3, h, w), np.float32)
        labels = -np.ones((batch, 1, 10), np.float32)
        for i in range(batch):
            if np.random.rand() > add_rate:
                continue
            plate, lp_type, _ = self.draw_LP()
            mask[i], fg[i], labels[i, 0, :9] = self.random_projection_LP_6D(plate, (h, w), r_max)
            labels[i, 0, 9] = lp_type
        return fg, mask, labels

    def add(self, bg_batch, r_max, add_rate=1.0):
        """(:134-166) bg_batch (B,3,h,w) float32 0..1 CUDA tensor (RenderCar.render's output) -> (images with plates, labels)."""
        import torch
        B, _, h, w = bg_batch.shape
        fg, mask, labels = self.add_host(B, h, w, r_max, add_rate)
        dev = bg_batch.device
        img = composite(bg_batch, torch.from_numpy(fg).to(dev), torch.from_numpy(mask).to(dev), unit_bg=True)
        return img, torch.from_numpy(labels).to(dev)
