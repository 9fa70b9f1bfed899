"""MXNet `.params` container + gluon parameter order (CPU only)."""
import struct

import numpy as np
import pytest

from yolo_amd import mxparams as mp
from yolo_amd.spec import NetGraph, darknet53_spec

MICRO = {'layers': [1, 1, 1], 'channels': [8, 16, 32, 64], 'all_anchors': [[[0.1, 0.1], [0.2, 0.3]]] * 2,
         'slice_point': [1, 3, 5, 6, 8]}


def _handmade(path):
    """Bytes laid out by hand from the documented layout (not by write_params): V2 f32, V2 f16, V1 i32, legacy f32."""
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    h = np.asarray([1.5, -2.25], np.float16)
    i = np.asarray([[7], [9]], np.int32)
    lg = np.asarray([0.5, 0.25, 0.125], np.float32)
    b = struct.pack('<QQQ', 0x112, 0, 4)
    b += struct.pack('<IiI', 0xF993FAC9, 0, 2) + struct.pack('<qq', 2, 3) + struct.pack('<iii', 2, 0, 0) + a.tobytes()
    b += struct.pack('<IiI', 0xF993FAC9, 0, 1) + struct.pack('<q', 2) + struct.pack('<iii', 1, 0, 2) + h.tobytes()
    b += struct.pack('<II', 0xF993FAC8, 2) + struct.pack('<qq', 2, 1) + struct.pack('<iii', 1, 0, 4) + i.tobytes()
    b += struct.pack('<I', 1) + struct.pack('<I', 3) + struct.pack('<iii', 1, 0, 0) + lg.tobytes()
    b += struct.pack('<Q', 4)
    for n in (b'net0_conv0_weight', b'net0_batchnorm0_gamma', b'idx', b'arg:legacy_bias'):
        b += struct.pack('<Q', len(n)) + n
    with open(path, 'wb') as f:
        f.write(b)
    return a, h, i, lg


def test_reader_against_handmade_bytes(tmp_path):
    p = str(tmp_path / 'hand.params')
    a, h, i, lg = _handmade(p)
    d = mp.read_params(p)
    assert list(d) == ['net0_conv0_weight', 'net0_batchnorm0_gamma', 'idx', 'arg:legacy_bias']
    np.testing.assert_array_equal(d['net0_conv0_weight'], a)
    assert d['net0_batchnorm0_gamma'].dtype == np.float16
    np.testing.assert_array_equal(d['net0_batchnorm0_gamma'], h)
    np.testing.assert_array_equal(d['idx'], i)
    np.testing.assert_array_equal(d['arg:legacy_bias'], lg)


def test_writer_reader_round_trip_and_layout(tmp_path):
    p = str(tmp_path / 'rt.params')
    src = {'w': np.random.default_rng(0).standard_normal((4, 3, 3, 3)).astype(np.float32), 'g': np.ones(4, np.float32)}
    mp.write_params(p, src)
    raw = open(p, 'rb').read()
    assert struct.unpack_from('<QQQ', raw, 0) == (0x112, 0, 2)
    assert struct.unpack_from('<IiI', raw, 24) == (0xF993FAC9, 0, 4)
    assert struct.unpack_from('<4q', raw, 36) == (4, 3, 3, 3)
    d = mp.read_params(p)
    assert list(d) == ['w', 'g']
    for k in src:
        np.testing.assert_array_equal(d[k], src[k])


def test_bad_files_are_rejected(tmp_path):
    p = str(tmp_path / 'bad.params')
    open(p, 'wb').write(struct.pack('<QQQ', 0x113, 0, 0))
    with pytest.raises(mp.ParamsFormatError):
        mp.read_params(p)
    open(p, 'wb').write(struct.pack('<QQQ', 0x112, 0, 1) + struct.pack('<IiI', 0xF993FAC9, 0, 1) + struct.pack('<q', 100))
    with pytest.raises(mp.ParamsFormatError):
        mp.read_params(p)


def _random_params(g, seed=0):
    rng = np.random.default_rng(seed)
    P = {}
    for c in g.convs():
        P[c.name + '.weight'] = rng.standard_normal((c.cout, c.cin, c.k, c.k)).astype(np.float32)
        if c.bn:
            for s in ('gamma', 'beta', 'running_mean', 'running_var'):
                P[c.name + '.' + s] = rng.standard_normal(c.cout).astype(np.float32)
        else:
            P[c.name + '.bias'] = rng.standard_normal(c.cout).astype(np.float32)
    return P


def test_gluon_order_registration_vs_forward():
    g = NetGraph(darknet53_spec())
    reg = [c.name for c in mp.gluon_conv_order(g, 'registration')]
    fwd = [c.name for c in mp.gluon_conv_order(g, 'forward')]
    assert reg[0] == fwd[0] == 'stem' and sorted(reg) == sorted(fwd) == sorted(c.name for c in g.convs())
    n_backbone = 1 + sum(1 + 2 * n for n in darknet53_spec()['layers'])
    assert reg[:n_backbone] == fwd[:n_backbone]
    # basic_yolo.py:31-36 registers transitions before yolo_blocks before yolo_outputs
    assert reg[n_backbone:n_backbone + 2] == ['transitions.0', 'transitions.1']
    assert reg[-3:] == ['heads.0.out', 'heads.1.out', 'heads.2.out']
    # the forward pass runs block 0, its output, then transition 0 (car/utils.py:76-93)
    assert fwd[n_backbone + 6:n_backbone + 8] == ['heads.0.out', 'transitions.0']


MID = {'layers': [1, 2, 1, 1], 'channels': [8, 16, 32, 64, 128], 'all_anchors': [[[0.1, 0.1]] * 3] * 3,
       'slice_point': [1, 3, 5, 6, 30]}


@pytest.mark.parametrize('spec', [MICRO, MID])
def test_gluon_file_round_trip(tmp_path, spec):
    g = NetGraph(spec)
    P = _random_params(g)
    p = str(tmp_path / 'net.params')
    mp.write_params(p, mp.to_gluon(g, P))
    back = mp.from_gluon(g, mp.read_params(p))
    assert sorted(back) == sorted(P)
    for k in P:
        np.testing.assert_array_equal(back[k], P[k])


def test_exported_symbol_order_and_shape_check(tmp_path):
    g = NetGraph(MICRO)
    P = _random_params(g, 1)
    exp = {}
    for c in mp.gluon_conv_order(g, 'forward'):          # net.export: arg:/aux: names, forward order
        exp['arg:%s_weight' % c.name] = P[c.name + '.weight']
        if c.bn:
            exp['arg:%s_gamma' % c.name] = P[c.name + '.gamma']
            exp['arg:%s_beta' % c.name] = P[c.name + '.beta']
        else:
            exp['arg:%s_bias' % c.name] = P[c.name + '.bias']
    for c in mp.gluon_conv_order(g, 'forward'):
        if c.bn:
            exp['aux:%s_running_mean' % c.name] = P[c.name + '.running_mean']
            exp['aux:%s_running_var' % c.name] = P[c.name + '.running_var']
    p = str(tmp_path / 'sym-0000.params')
    mp.write_params(p, exp)
    back = mp.from_gluon(g, mp.read_params(p))
    for k in P:
        np.testing.assert_array_equal(back[k], P[k])
    # a file in the other order fails the shape check instead of loading silently wrong
    with pytest.raises(mp.ParamsFormatError):
        mp.from_gluon(g, mp.read_params(p), order='registration')


def test_gluon_order_with_lp_branch():
    spec = dict(MID, LP_slice_point=[1, 3, 4, 7, 10])
    g = NetGraph(spec)
    reg = [c.name for c in mp.gluon_conv_order(g, 'registration')]
    fwd = [c.name for c in mp.gluon_conv_order(g, 'forward')]
    assert sorted(reg) == sorted(fwd) == sorted(c.name for c in g.convs()) and len(reg) == len(set(reg))
    # CarLPNet.__init__ (car_and_LP/YOLO.py:47-60) registers LP_branch after the base class's blocks ...
    assert reg[-31:] == ['lp.%d.%s' % (k, n) for k in range(5) for n in ('b0', 'b1', 'b2', 'b3', 'b4', 'tip')] + ['lp.out']
    # ... and runs it before the finest detection block (:72-79)
    assert fwd.index('lp.out') + 1 == fwd.index('heads.2.b0')
    P = _random_params(g, 2)
    back = mp.from_gluon(g, mp.to_gluon(g, P))
    assert all(np.array_equal(back[k], P[k]) for k in P)
