"""Oracle: spec -> layer graph and seeded parameters (test infrastructure only).

Restates
  * BasicYOLONet.__init__            yolo_modules/basic_yolo.py:8-39
  * YOLOOutput                       yolo_modules/basic_yolo.py:91-105
  * YOLOPyrmaid                      yolo_modules/basic_yolo.py:108-123
  * gluoncv 0.4.0b20181129 blocks the reference imports (basic_yolo.py:3-4),
    absent from /root/reference, restated from their published definitions:
      _conv2d(channel, kernel, padding, stride)  = Conv2D(no bias) + BatchNorm(eps 1e-5) + LeakyReLU(0.1)
      DarknetBasicBlockV3(channel)               = x + conv3x3(2*channel)(conv1x1(channel)(x))
      YOLODetectionBlockV3(channel)              = body[1x1 c,3x3 2c,1x1 c,3x3 2c,1x1 c] -> route ; tip = 3x3 2c(route)
  * mxnet.init.Xavier() defaults (yolo_gluon.py:198): uniform, factor_type avg, magnitude 3.

PARITY UNPINNED (see oracle/__init__.py).
"""
import numpy as np

BN_EPS = 1e-5
LEAKY = 0.1


def _conv(name, cin, cout, k, stride, bn=True):
    return dict(name=name, cin=int(cin), cout=int(cout), k=int(k), stride=int(stride),
                pad=(k // 2), bn=bn)


def build_graph(spec, in_channels=3):
    """basic_yolo.py:8-39.  Returns dict(stem, stages, heads, transitions, num_pyramid)."""
    layers = spec['layers']
    channels = spec['channels']
    assert len(layers) == len(channels) - 1          # basic_yolo.py:14
    g = {}
    g['stem'] = _conv('stem', in_channels, channels[0], 3, 1)       # :20
    stages = []
    prev = channels[0]
    for i, (nlayer, channel) in enumerate(zip(layers, channels[1:])):   # :22
        st = dict(down=_conv('stages.%d.down' % i, prev, channel, 3, 2), res=[])   # :24
        for j in range(nlayer):                                      # :25-26
            half = channel // 2
            st['res'].append((
                _conv('stages.%d.res.%d.c1' % (i, j), channel, half, 1, 1),
                _conv('stages.%d.res.%d.c2' % (i, j), half, half * 2, 3, 1)))
        stages.append(st)
        prev = channel
    g['stages'] = stages

    anchors = spec['all_anchors']                                    # :29
    n_pyr = len(anchors)
    per_anchor = spec['slice_point'][-1]                             # :35
    pyr = channels[-n_pyr:]                                          # :33
    heads, transitions = [], []
    # YOLOPyrmaid, basic_yolo.py:108-123: deep -> shallow, anchors reversed
    for i, (channel, anchor) in enumerate(zip(pyr[::-1], anchors[::-1])):
        cin = channels[-1] if i == 0 else 2 * channel   # concat(upsampled transition, route)
        body = []
        c_prev = cin
        for b, (cout, k) in enumerate([(channel, 1), (2 * channel, 3), (channel, 1),
                                       (2 * channel, 3), (channel, 1)]):
            body.append(_conv('heads.%d.b%d' % (i, b), c_prev, cout, k, 1))
            c_prev = cout
        tip = _conv('heads.%d.tip' % i, channel, 2 * channel, 3, 1)
        out = _conv('heads.%d.out' % i, 2 * channel, per_anchor * len(anchor), 1, 1, bn=False)  # :98
        heads.append(dict(body=body, tip=tip, out=out, num_anchors=len(anchor)))
        if i > 0:                                                    # :120-121
            transitions.append(_conv('transitions.%d' % (i - 1), pyr[::-1][i - 1], channel, 1, 1))
    g['heads'] = heads
    g['transitions'] = transitions
    if 'LP_slice_point' in spec:
        # CarLPNet.__init__, car_and_LP/YOLO.py:47-60: five YOLODetectionBlockV3(channels[-3]) chained through their
        # TIP outputs on the input of the finest detection block, then a biased 1x1 to LP_slice_point[-1] channels
        lpc = channels[-3]
        blocks, c_prev = [], 2 * pyr[0] if n_pyr > 1 else channels[-1]
        for k in range(5):
            body = []
            for b, (cout, kk) in enumerate([(lpc, 1), (2 * lpc, 3), (lpc, 1), (2 * lpc, 3), (lpc, 1)]):
                body.append(_conv('lp.%d.b%d' % (k, b), c_prev, cout, kk, 1))
                c_prev = cout
            tip = _conv('lp.%d.tip' % k, lpc, 2 * lpc, 3, 1)
            blocks.append(dict(body=body, tip=tip))
            c_prev = 2 * lpc
        g['lp'] = dict(blocks=blocks, out=_conv('lp.out', 2 * lpc, spec['LP_slice_point'][-1], 1, 1, bn=False))
    g['num_pyramid'] = n_pyr
    g['per_anchor'] = per_anchor
    return g


def conv_list(g):
    """Deterministic parameter order: stem, stages, heads (deep->shallow), transitions, LP branch."""
    out = [g['stem']]
    for st in g['stages']:
        out.append(st['down'])
        for c1, c2 in st['res']:
            out += [c1, c2]
    for h in g['heads']:
        out += h['body'] + [h['tip'], h['out']]
    out += g['transitions']
    if 'lp' in g:                      # registered last (CarLPNet.__init__ adds LP_branch after the base class)
        for blk in g['lp']['blocks']:
            out += blk['body'] + [blk['tip']]
        out.append(g['lp']['out'])
    return out


def init_params(g, seed=0, bn='identity'):
    """Xavier-uniform weights (mxnet.init.Xavier defaults) from numpy default_rng(seed),
    drawn in conv_list order.  bn='identity': gamma 1, beta 0, mean 0, var 1 (MXNet
    initial state); bn='random': non-trivial statistics so BN folding is exercised."""
    rng = np.random.default_rng(seed)
    P = {}
    for c in conv_list(g):
        k2 = c['k'] * c['k']
        fan_in, fan_out = c['cin'] * k2, c['cout'] * k2
        a = np.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
        P[c['name'] + '.weight'] = rng.uniform(-a, a, (c['cout'], c['cin'], c['k'], c['k'])).astype(np.float32)
        n = c['cout']
        if c['bn']:
            if bn == 'identity':
                P[c['name'] + '.gamma'] = np.ones(n, np.float32)
                P[c['name'] + '.beta'] = np.zeros(n, np.float32)
                P[c['name'] + '.running_mean'] = np.zeros(n, np.float32)
                P[c['name'] + '.running_var'] = np.ones(n, np.float32)
            else:
                P[c['name'] + '.gamma'] = rng.uniform(0.5, 1.5, n).astype(np.float32)
                P[c['name'] + '.beta'] = (0.1 * rng.standard_normal(n)).astype(np.float32)
                P[c['name'] + '.running_mean'] = (0.1 * rng.standard_normal(n)).astype(np.float32)
                P[c['name'] + '.running_var'] = rng.uniform(0.5, 1.5, n).astype(np.float32)
        else:
            if bn == 'identity':
                P[c['name'] + '.bias'] = np.zeros(n, np.float32)
            else:
                P[c['name'] + '.bias'] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    return P


def count_params(g):
    n = 0
    for c in conv_list(g):
        n += c['cout'] * c['cin'] * c['k'] * c['k']
        n += 2 * c['cout'] if c['bn'] else c['cout']      # gamma,beta | bias (trainable only)
    return n


def conv_flops(g, h, w):
    """2*Cin*k^2*Cout*Hout*Wout summed over convs, following the forward's spatial sizes."""
    total = 0
    def f(c, hh, ww):
        ho, wo = (hh + 2 * c['pad'] - c['k']) // c['stride'] + 1, (ww + 2 * c['pad'] - c['k']) // c['stride'] + 1
        return 2 * c['cin'] * c['k'] ** 2 * c['cout'] * ho * wo, ho, wo
    fl, h, w = f(g['stem'], h, w); total += fl
    sizes = []
    for st in g['stages']:
        fl, h, w = f(st['down'], h, w); total += fl
        for c1, c2 in st['res']:
            total += f(c1, h, w)[0] + f(c2, h, w)[0]
        sizes.append((h, w))
    sizes = sizes[-g['num_pyramid']:][::-1]
    for i, hd in enumerate(g['heads']):
        hh, ww = sizes[i]
        for c in hd['body'] + [hd['tip'], hd['out']]:
            total += f(c, hh, ww)[0]
        if i < len(g['transitions']):
            total += f(g['transitions'][i], hh, ww)[0]
    return total


# Specs in the reference's own schema -------------------------------------------------
CAR_ANCHORS = [[[0.2216, 0.1552], [0.2144, 0.2408], [0.2825, 0.3456]],
               [[0.3959, 0.2706], [0.3703, 0.4351], [0.5708, 0.4278]],
               [[0.4345, 0.6063], [0.5584, 0.7174], [0.7448, 0.6772]]]   # car/v1/spec.yaml:7-11

def spec_d53():
    """Canonical Darknet-53 in the reference's spec format; car head C=30."""
    return dict(layers=[1, 2, 8, 8, 4], channels=[32, 64, 128, 256, 512, 1024],
                slice_point=[1, 3, 5, 6, 30], all_anchors=CAR_ANCHORS)

def spec_car_v1():
    """car/v1/spec.yaml:4-6 (native size 320x512)."""
    return dict(layers=[1, 4, 4, 8, 8, 4], channels=[16, 32, 64, 128, 256, 512, 1024],
                slice_point=[1, 3, 5, 6, 30], all_anchors=CAR_ANCHORS)

def spec_test_yaml():
    """yolo_modules/test.yaml:1-12."""
    return dict(layers=[1, 2, 2, 4, 4, 4], channels=[8, 16, 32, 64, 128, 256, 512],
                slice_point=[1, 5, 17],
                all_anchors=[[[0.3, 0.3], [0.4, 0.2], [0.2, 0.4]],
                             [[0.5, 0.5], [0.4, 0.6], [0.6, 0.4]],
                             [[0.7, 0.7], [0.8, 0.6], [0.6, 0.8]]])

def spec_micro():
    """Small net in the reference's schema for the fp64 loop oracle and fast GPU tests."""
    return dict(layers=[1, 1, 2, 1, 1], channels=[8, 16, 32, 64, 64, 128],
                slice_point=[1, 3, 5, 6, 10], all_anchors=CAR_ANCHORS)
