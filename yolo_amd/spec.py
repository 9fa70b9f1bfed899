"""Spec-driven layer graph: the host-side mirror of BasicYOLONet.__init__
(yolo_modules/basic_yolo.py:8-39) and YOLOPyrmaid (:108-123).

The spec dict is the reference's own model-definition API (<version>/spec.yaml keys `layers`,
`channels`, `all_anchors`, `slice_point`; car/v1/spec.yaml:1-11), so any reference spec builds here.
"""
import math

BN_EPS = 1e-5          # gluoncv _conv2d: BatchNorm(epsilon=1e-5, momentum=0.9)
BN_MOMENTUM = 0.9
LEAKY_SLOPE = 0.1      # gluoncv _conv2d: LeakyReLU(0.1)


class ConvSpec(object):
    """One gluoncv `_conv2d` (conv + BN + LeakyReLU) or, with bn=False, YOLOOutput's biased 1x1."""
    __slots__ = ('name', 'cin', 'cout', 'k', 'stride', 'bn')

    def __init__(self, name, cin, cout, k, stride=1, bn=True):
        self.name, self.cin, self.cout, self.k, self.stride, self.bn = name, int(cin), int(cout), int(k), int(stride), bn

    @property
    def pad(self):
        return self.k // 2

    def out_hw(self, h, w):
        return ((h + 2 * self.pad - self.k) // self.stride + 1, (w + 2 * self.pad - self.k) // self.stride + 1)

    def param_names(self):
        if self.bn:
            return [self.name + s for s in ('.weight', '.gamma', '.beta', '.running_mean', '.running_var')]
        return [self.name + '.weight', self.name + '.bias']


class NetGraph(object):
    def __init__(self, spec, in_channels=3):
        layers, channels = spec['layers'], spec['channels']
        if len(layers) != len(channels) - 1:
            raise ValueError('len(channels) should equal to len(layers) + 1, given {} vs {}'.format(
                len(channels), len(layers)))                      # basic_yolo.py:14-16
        self.spec = spec
        self.anchors = spec['all_anchors']
        self.slice_point = list(spec['slice_point'])
        self.per_anchor = self.slice_point[-1]
        self.num_pyramid = len(self.anchors)
        self.stem = ConvSpec('stem', in_channels, channels[0], 3, 1)
        self.stages = []
        prev = channels[0]
        for i, (n, ch) in enumerate(zip(layers, channels[1:])):
            down = ConvSpec('stages.%d.down' % i, prev, ch, 3, 2)
            res = [(ConvSpec('stages.%d.res.%d.c1' % (i, j), ch, ch // 2, 1),
                    ConvSpec('stages.%d.res.%d.c2' % (i, j), ch // 2, (ch // 2) * 2, 3)) for j in range(n)]
            self.stages.append((down, res))
            prev = ch
        pyr = channels[-self.num_pyramid:][::-1]
        self.heads, self.transitions = [], []
        for i, (ch, anchor) in enumerate(zip(pyr, self.anchors[::-1])):
            cin = channels[-1] if i == 0 else 2 * ch
            body, c_prev = [], cin
            for b, (cout, k) in enumerate([(ch, 1), (2 * ch, 3), (ch, 1), (2 * ch, 3), (ch, 1)]):
                body.append(ConvSpec('heads.%d.b%d' % (i, b), c_prev, cout, k))
                c_prev = cout
            tip = ConvSpec('heads.%d.tip' % i, ch, 2 * ch, 3)
            out = ConvSpec('heads.%d.out' % i, 2 * ch, self.per_anchor * len(anchor), 1, bn=False)
            self.heads.append((body, tip, out, len(anchor)))
            if i > 0:
                self.transitions.append(ConvSpec('transitions.%d' % (i - 1), pyr[i - 1], ch, 1))
        # CarLPNet's licence-plate branch (car_and_LP/YOLO.py:47-60): five YOLODetectionBlockV3(channels[-3]) chained
        # through their tip outputs on the input of the finest detection block + a biased 1x1 to LP_slice_point[-1]
        self.lp_blocks, self.lp_out = [], None
        if 'LP_slice_point' in spec:
            self.lp_slice_point = list(spec['LP_slice_point'])
            lpc = channels[-3]
            c_prev = 2 * pyr[-1] if self.num_pyramid > 1 else channels[-1]
            for k in range(5):
                body = []
                for b, (cout, kk) in enumerate([(lpc, 1), (2 * lpc, 3), (lpc, 1), (2 * lpc, 3), (lpc, 1)]):
                    body.append(ConvSpec('lp.%d.b%d' % (k, b), c_prev, cout, kk))
                    c_prev = cout
                self.lp_blocks.append((body, ConvSpec('lp.%d.tip' % k, lpc, 2 * lpc, 3)))
                c_prev = 2 * lpc
            self.lp_out = ConvSpec('lp.out', 2 * lpc, self.lp_slice_point[-1], 1, bn=False)

    def convs(self):
        out = [self.stem]
        for down, res in self.stages:
            out.append(down)
            for c1, c2 in res:
                out += [c1, c2]
        for body, tip, o, _ in self.heads:
            out += body + [tip, o]
        out = out + self.transitions
        for body, tip in self.lp_blocks:
            out += body + [tip]
        return out + ([self.lp_out] if self.lp_out is not None else [])

    def steps(self):
        """car/YOLO.py:112-116."""
        nd, npy = len(self.spec['layers']), self.num_pyramid
        return [2 ** (nd - npy + 1 + i) for i in range(npy)]

    def flops(self, h, w):
        tot = 0
        def f(c, hh, ww):
            ho, wo = c.out_hw(hh, ww)
            return 2 * c.cin * c.k * c.k * c.cout * ho * wo, ho, wo
        fl, h, w = f(self.stem, h, w); tot += fl
        sizes = []
        for down, res in self.stages:
            fl, h, w = f(down, h, w); tot += fl
            for c1, c2 in res:
                tot += f(c1, h, w)[0] + f(c2, h, w)[0]
            sizes.append((h, w))
        sizes = sizes[-self.num_pyramid:][::-1]
        for i, (body, tip, o, _) in enumerate(self.heads):
            hh, ww = sizes[i]
            for c in body + [tip, o]:
                tot += f(c, hh, ww)[0]
            if i < len(self.transitions):
                tot += f(self.transitions[i], hh, ww)[0]
        if self.lp_out is not None:
            hh, ww = sizes[-1]
            for body, tip in self.lp_blocks:
                for c in body + [tip]:
                    tot += f(c, hh, ww)[0]
            tot += f(self.lp_out, hh, ww)[0]
        return tot


CAR_ANCHORS = [[[0.2216, 0.1552], [0.2144, 0.2408], [0.2825, 0.3456]],
               [[0.3959, 0.2706], [0.3703, 0.4351], [0.5708, 0.4278]],
               [[0.4345, 0.6063], [0.5584, 0.7174], [0.7448, 0.6772]]]      # car/v1/spec.yaml:7-11


def darknet53_spec():
    """Canonical Darknet-53 expressed in the reference's spec format (BASELINE.json configs 2-5)."""
    return dict(layers=[1, 2, 8, 8, 4], channels=[32, 64, 128, 256, 512, 1024],
                slice_point=[1, 3, 5, 6, 30], all_anchors=CAR_ANCHORS)


def xavier_bound(cin, cout, k):
    """mxnet.init.Xavier() defaults (yolo_gluon.py:198): uniform, factor_type avg, magnitude 3."""
    return math.sqrt(3.0 / ((cin * k * k + cout * k * k) / 2.0))
