"""Oracle: the synthetic-target generators (test infrastructure only -- PARITY UNPINNED, see oracle/__init__.py).

Restates, as plain functions that replay the reference's sequence of random draws on PIL images,
  RenderCar.render / _render_png / _render_pascal / _resize      car/render_car.py:52-138, 262-392
  RenderCar.get_pascal3d_azi_ele                                 car/render_car.py:440-458
  PILImageEnhance (shear off; rotate, blur, noise)               yolo_modules/yolo_cv.py:95-157
  pil_rgb_2_rgb_ndarray / pil_mask_2_rgb_ndarray                 yolo_modules/yolo_gluon.py:298-313
  LPGenerator.draw_LP / random_projection_LP_6D / add            yolo_modules/licence_plate_render/__init__.py:58-166
  ProjectRectangle6D.__call__ / projection_matrix                yolo_modules/licence_plate_render/__init__.py:273-371
  cv2.getPerspectiveTransform (cv2 is absent: its published definition -- the 3x3 homography with h33 = 1 through four
  point pairs -- solved as the 8x8 linear system)
mxnet's colour augmenters are NOT restated a second time: the comparisons run with the augmenters off, and the augmenter of
the product has its own known-answer test (tests/test_render.py).

Written from the reference sources, independently of yolo_amd/render.py: geometry is measured on the images (numpy on the
alpha channel) where the product trusts PIL's getbbox, and the blend is evaluated in float64.
"""
import math

import numpy as np
from PIL import Image, ImageFilter

from .train import get_label_dist


# ---- yolo_cv.PILImageEnhance --------------------------------------------------------------------------------------------
def enhance(img, R=0.0, G=0.0, noise_var=0.0):
    """yolo_cv.py:103-157 with M = N = 0: [rotate by U(-R, R) degrees, expand] -> [Gaussian blur, radius rand() * G] ->
    [N(0, noise_var) noise on every channel, clip, truncate to uint8].  Returns (image, r in radians)."""
    r = 0
    if R != 0:
        deg = np.random.uniform(low=-R, high=R)
        img = img.rotate(deg, Image.BILINEAR, expand=1)
        r = float(deg * np.pi) / 180
    if G != 0:
        img = img.filter(ImageFilter.GaussianBlur(radius=np.random.rand() * G))
    if noise_var != 0:
        a = np.array(img)
        a = np.clip(a + np.random.normal(0., noise_var, a.shape), 0, 255)
        img = Image.fromarray(np.uint8(a))
    return img, r


def alpha_bbox(img):
    """(l, t, r, b) of the non-transparent pixels -- PIL's getbbox() of an RGBA image is the box of pixels that are non-zero
    in ANY band, which for sprites on a (0,0,0,0) canvas is the alpha box unless a transparent pixel carries colour."""
    a = np.asarray(img)
    nz = np.nonzero(a.reshape(a.shape[0], a.shape[1], -1).any(axis=2))
    return int(nz[1].min()), int(nz[0].min()), int(nz[1].max()) + 1, int(nz[0].max()) + 1


# ---- RenderCar ----------------------------------------------------------------------------------------------------------
def pascal_azi_ele(mat):
    """render_car.py:440-458 on a loaded PASCAL3D+ annotation (scipy.io.loadmat dict): (ele, azi [rad], box, skip)."""
    objs = mat['record'][0][0][1][0]
    if len(objs) > 1:
        return 0, 0, 0, True
    box = [int(i) for i in objs[0][1][0]]
    ele = float(np.ravel(objs[0][3][0][0][3][0])[0]) * math.pi / 180.
    azi = float(np.ravel(objs[0][3][0][0][2][0])[0]) * math.pi / 180.
    return ele, azi, box, False


def render_batch(bg, png_paths, pascal_set, classes, img_h, img_w, pascal_rate=0.0, render_rate=1.0, R=30.0, G=0.3):
    """RenderCar.render (render_car.py:52-138).  bg (B,3,H,W) 0..255; png_paths: the mode's sprite paths in listing order;
    pascal_set: [(RGBA image, box, cls, dist)] as load_pascal_images pre-loads them (render_car.py:243-252).
    Returns (images float64 0..1, labels float32 (B,1,6+ncls))."""
    B = len(bg)
    ncls = len(classes)
    fg = np.zeros((B, 3, img_h, img_w))
    mask = np.zeros((B, 3, img_h, img_w))
    labels = -np.ones((B, 1, 6 + ncls), np.float32)
    for i in range(B):
        if np.random.rand() > render_rate:
            continue
        r1 = np.random.uniform(low=0.9, high=1.1)
        if np.random.rand() < pascal_rate:
            # ---- _render_pascal (render_car.py:262-337) ------------------------------------------------------
            n = np.random.randint(len(pascal_set))
            sprite, box, cls, dist = pascal_set[n]
            bl, bt, br, bb = box
            bw, bh = br - bl, (bb - bt) * r1
            hi = min(0.9 * img_w / bw, 0.9 * img_h / bh)
            lo = max(0.2 * img_w / float(bw), 0.2 * img_h / float(bh))
            s = np.random.uniform(low=lo, high=hi)
            rw, rh = s * sprite.size[0], s * sprite.size[1] * r1
            sprite = sprite.resize((int(rw), int(rh)), Image.BILINEAR)
            # pil_image_enhance(pil_img, R=0) (render_car.py:306): the enhancer's own R is 30, so random_rotate RUNS -- with
            # the keyword's R = 0: one uniform(-0, 0) draw, a rotation by 0 degrees, r = 0 -- then the blur
            deg = np.random.uniform(low=-0.0, high=0.0)
            sprite = sprite.rotate(deg, Image.BILINEAR, expand=1)
            r = float(deg * np.pi) / 180
            sprite, _ = enhance(sprite, R=0.0, G=G)
            # the annotated box, centred, turned by r with the image, shifted to the expanded canvas
            xs = [bl * s - 0.5 * rw, br * s - 0.5 * rw]
            ys = [bt * s * r1 - 0.5 * rh, bb * s * r1 - 0.5 * rh]
            pts = np.array([[x * math.cos(r) - y * math.sin(r), y * math.cos(r) + x * math.sin(r)] for x in xs for y in ys])
            off = 0.5 * np.array([abs(rh * math.sin(r)) + abs(rw * math.cos(r)), abs(rh * math.cos(r)) + abs(rw * math.sin(r))])
            (L, T), (Rr, Bb) = pts.min(axis=0) + off, pts.max(axis=0) + off
        else:
            # ---- _render_png (render_car.py:339-368) ---------------------------------------------------------
            n = np.random.randint(len(png_paths))
            name = png_paths[n].split('/')[-1]
            ele = float(name.split('ele')[1].split('.')[0]) * math.pi / 18000.
            azi = float(name.split('azi')[1].split('_')[0]) * math.pi / 18000.
            cls, dist = get_label_dist(ele, azi, classes)
            sprite = Image.open(png_paths[n]).convert('RGBA')
            s = np.random.uniform(low=0.2, high=1.0)
            sprite = sprite.resize((int(s * sprite.size[0]), int(s * sprite.size[1] * r1)), Image.BILINEAR)
            sprite, r = enhance(sprite, R=R, G=G)
            L, T, Rr, Bb = alpha_bbox(sprite)
        bw, bh = Rr - L, Bb - T
        px = np.random.randint(low=int(-L - 0.3 * bw), high=int(img_w - L - 0.7 * bw))
        py = np.random.randint(low=int(-T - 0.3 * bh), high=int(img_h - T - 0.7 * bh))
        canvas = Image.new('RGBA', (img_w, img_h))
        canvas.paste(sprite, (px, py))
        a = np.asarray(canvas, np.float64)
        fg[i] = a[..., :3].transpose(2, 0, 1) / 255.
        mask[i] = a[..., 3][None] / 255.
        labels[i, 0, :6] = [cls, ((Bb + T) / 2. + py) / img_h, ((Rr + L) / 2. + px) / img_w, float(Bb - T) / img_h,
                            float(Rr - L) / img_w, r]
        labels[i, 0, 6:] = np.asarray(dist, np.float32).reshape(-1)
    img = np.clip((np.asarray(bg, np.float64) / 255.) * (1 - mask) + fg * mask, 0, 1)
    return img, labels


# ---- LPGenerator ----------------------------------------------------------------------------------------------------------
def perspective_through(src, dst):
    """cv2.getPerspectiveTransform(src, dst): the 3x3 matrix M, M[2,2] = 1, with dst ~ M @ [x, y, 1] for the four pairs."""
    A, b = np.zeros((8, 8)), np.zeros(8)
    for k, ((x, y), (u, v)) in enumerate(zip(np.asarray(src, np.float64), np.asarray(dst, np.float64))):
        A[k] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        A[k + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[k], b[k + 4] = u, v
    h = np.linalg.solve(A, b)
    return np.append(h, 1.0).reshape(3, 3)


def project_plate(pose, cam):
    """ProjectRectangle6D.__call__ (licence_plate_render/__init__.py:317-371): the four plate corners in camera pixels for
    pose [X, Y, Z (mm), r1, r2, r3 (rad)]; the half sizes 199.5 x 84.0 are the reference's constants."""
    X, Y, Z, r1, r2, r3 = pose
    fx, fy, cx, cy = cam['fx'], cam['fy'], cam['cx'], cam['cy']
    s, c = math.sin, math.cos
    a, b_ = s(r1) * c(r2) * 84.0, s(r1) * s(r2) * c(r3) * 84.0
    cc, d = s(r2) * 199.5, s(r3) * c(r1) * 84.0
    e, f = c(r2) * c(r3) * 199.5, s(r1) * s(r2) * s(r3) * 84.0
    g, h = s(r3) * c(r2) * 199.5, c(r1) * c(r3) * 84.0
    den = [Z + a - cc, Z + a + cc, Z - a + cc, Z - a - cc]
    xn = [cx * den[0] + fx * (X + b_ - d + e), cx * den[1] + fx * (X + b_ - d - e), cx * den[2] + fx * (X - b_ + d - e),
          cx * den[3] + fx * (X - b_ + d + e)]
    yn = [cy * den[0] + fy * (Y + f + g + h), cy * den[1] + fy * (Y + f - g + h), cy * den[2] + fy * (Y - f - g - h),
          cy * den[3] + fy * (Y - f + g - h)]
    return np.array([[xn[k] / den[k], yn[k] / den[k]] for k in range(4)], np.float32)


def draw_plate(font, dot, colour=(255, 255, 255)):
    """draw_LP (:58-77), type 0 'ABC-1234': three letters (glyph ids 10..33), the dot, four digits 0..8 with 4 -> 9."""
    xs = [7, 56, 106, 158, 175, 225, 274, 324]
    plate = Image.new('RGBA', (380, 160), colour)
    for k, j in enumerate(np.random.randint(10, 34, size=3)):
        plate.paste(font[j], (xs[k], 35))
    plate.paste(dot, (xs[3], 45))
    for k, j in enumerate(np.random.randint(0, 9, size=4)):
        plate.paste(font[9 if j == 4 else j], (xs[k + 4], 35))
    return plate


def add_plates(bg, r_max, font, dot, cam, add_rate=1.0):
    """LPGenerator.add (:134-166) with the colour augmenter off: bg (B,3,H,W) 0..1 -> (images float64, labels (B,1,10)
    [1, X, Y, Z, r1, r2, r3, x_px, y_px, type]; -1 rows: no plate)."""
    B, _, H, W = bg.shape
    fg, mask = np.zeros(bg.shape), np.zeros(bg.shape)
    labels = -np.ones((B, 1, 10), np.float32)
    for i in range(B):
        if np.random.rand() > add_rate:
            continue
        plate = draw_plate(font, dot)
        Z = np.random.uniform(low=1500., high=5000.)
        X = (Z * 9 / 30.) * np.random.uniform(low=-1, high=1)
        Y = (Z * 7 / 30.) * np.random.uniform(low=-1, high=1)
        r = [np.random.uniform(low=-1, high=1) * r_max[k] * math.pi / 180. for k in range(3)]
        pts = project_plate([X, Y, Z] + r, cam)
        M = perspective_through(pts, np.float32([[380, 160], [0, 160], [0, 0], [380, 0]]))
        plate = plate.transform((cam['w'], cam['h']), Image.PERSPECTIVE, tuple(M.reshape(-1)[:8]), Image.BILINEAR)
        plate = plate.resize((W, H), Image.BILINEAR)
        plate, _ = enhance(plate, G=1.0, noise_var=5.0)
        a = np.asarray(plate, np.float64)
        fg[i] = a[..., :3].transpose(2, 0, 1) / 255.
        mask[i] = a[..., 3][None] / 255.
        x = (X * cam['fx'] / Z + cam['cx']) * W / float(cam['w'])
        y = (Y * cam['fy'] / Z + cam['cy']) * H / float(cam['h'])
        labels[i, 0] = [1, X, Y, Z, r[0], r[1], r[2], x, y, 0]
    return np.clip(np.asarray(bg, np.float64) * (1 - mask) + fg * mask, 0, 1), labels
