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


def _handnamed(g, P, shuffle_seed=None, export=False):
    """A checkpoint dict with gluon's names written out BY HAND from the naming rules (not through
    mxparams.gluon_param_names): backbone under the net's prefix with running counters; YOLOPyrmaid's blocks outside
    every scope -- per scale the output conv (top-level conv counter), the detection block (its own scope), and for
    i > 0 the transition (top-level conv + batchnorm counters)."""
    items = []

    def add(name, ours):
        if export:
            name = ('aux:' if 'running' in name else 'arg:') + name
        items.append((name, P[ours]))

    n = 0
    for c in [g.stem] + [x for down, res in g.stages for x in [down] + [y for pair in res for y in pair]]:
        add('carnet0_conv%d_weight' % n, c.name + '.weight')
        for k in ('gamma', 'beta', 'running_mean', 'running_var'):
            add('carnet0_batchnorm%d_%s' % (n, k), c.name + '.' + k)
        n += 1
    top_c = top_b = 0
    for i, (body, tip, outc, _) in enumerate(g.heads):
        add('conv%d_weight' % top_c, outc.name + '.weight'); add('conv%d_bias' % top_c, outc.name + '.bias'); top_c += 1
        for jj, c in enumerate(list(body) + [tip]):
            add('yolodetectionblockv3%d_conv%d_weight' % (i, jj), c.name + '.weight')
            for k in ('gamma', 'beta', 'running_mean', 'running_var'):
                add('yolodetectionblockv3%d_batchnorm%d_%s' % (i, jj, k), c.name + '.' + k)
        if i > 0:
            t = g.transitions[i - 1]
            add('conv%d_weight' % top_c, t.name + '.weight'); top_c += 1
            for k in ('gamma', 'beta', 'running_mean', 'running_var'):
                add('batchnorm%d_%s' % (top_b, k), t.name + '.' + k)
            top_b += 1
    if shuffle_seed is not None:
        np.random.default_rng(shuffle_seed).shuffle(items)
    from collections import OrderedDict
    return OrderedDict(items)


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


@pytest.mark.parametrize('spec', [MICRO, MID])
def test_names_written_match_gluon_naming_rules(spec):
    """to_gluon's names == the names derived by hand from gluon's scoping rules, in collect_params() order for the
    backbone, and as a set overall."""
    g = NetGraph(spec)
    P = _random_params(g, 3)
    ours, hand = mp.to_gluon(g, P), _handnamed(g, P)
    assert set(ours) == set(hand)
    for k in hand:
        np.testing.assert_array_equal(ours[k], hand[k])
    nb = 5 * (1 + sum(1 + 2 * n for n in spec['layers']))
    assert list(ours)[:nb] == list(hand)[:nb]
    # registration order after the backbone: transitions, then the blocks, then the outputs (basic_yolo.py:32-37)
    rest = list(ours)[nb:]
    assert rest[0] == 'conv2_weight' and rest[1].startswith('batchnorm0_')
    assert rest[-2:] == ['conv%d_weight' % (3 if len(g.heads) == 3 else 1), 'conv%d_bias' % (3 if len(g.heads) == 3 else 1)]


@pytest.mark.parametrize('export', [False, True])
def test_real_files_are_in_hash_order_match_by_name(tmp_path, export):
    """The reference is Python-2 code and saves through a plain dict: the order inside a real file is arbitrary.  Many
    layers share a shape (repeated residual blocks, the per-conv gamma/beta), so a positional mapping would permute
    them silently; matching by name must not care."""
    g = NetGraph(MID)
    P = _random_params(g, 4)
    for seed in (0, 1, 2):
        p = str(tmp_path / ('s%d.params' % seed))
        mp.write_params(p, _handnamed(g, P, shuffle_seed=seed, export=export))
        back = mp.from_gluon(g, mp.read_params(p))
        assert sorted(back) == sorted(P)
        for k in P:
            np.testing.assert_array_equal(back[k], P[k], err_msg=k)


def test_name_matching_ignores_counter_offsets_and_prefix(tmp_path):
    """A net that was not the first block of its process has shifted counters (carnet3_, conv7_ ...): only the structure
    (scope, kind, counter ORDER) is used."""
    g = NetGraph(MICRO)
    P = _random_params(g, 5)
    import re
    shifted = {}
    for n, a in _handnamed(g, P).items():
        n = n.replace('carnet0_', 'carnet3_')
        n = re.sub(r'^(conv|batchnorm)(\d+)_', lambda m: '%s%d_' % (m.group(1), int(m.group(2)) + 7), n)
        n = re.sub(r'yolodetectionblockv3(\d+)_', lambda m: 'yolodetectionblockv3%d_' % (int(m.group(1)) + 11), n)
        shifted[n] = a
    back = mp.from_gluon(g, shifted)
    for k in P:
        np.testing.assert_array_equal(back[k], P[k], err_msg=k)


def test_wrong_structure_is_rejected():
    g = NetGraph(MICRO)
    P = _random_params(g, 6)
    d = _handnamed(g, P)
    missing = dict(d); del missing['carnet0_batchnorm3_gamma']
    with pytest.raises(mp.ParamsFormatError):
        mp.from_gluon(g, missing)
    other = _handnamed(NetGraph(MID), _random_params(NetGraph(MID), 6))
    with pytest.raises(mp.ParamsFormatError):
        mp.from_gluon(g, other)


def test_positional_fallback_for_files_without_gluon_names(tmp_path):
    """Name-less containers ('0', '1', ...) or foreign names: registration order, every shape checked."""
    g = NetGraph(MICRO)
    P = _random_params(g, 1)
    exp = {}
    for c in mp.gluon_conv_order(g, 'registration'):
        for k in (('weight', 'gamma', 'beta', 'running_mean', 'running_var') if c.bn else ('weight', 'bias')):
            exp['%s_%s' % (c.name, k)] = P[c.name + '.' + k]
    back = mp.from_gluon(g, exp)
    for k in P:
        np.testing.assert_array_equal(back[k], P[k])
    # a file in the other order fails the shape check instead of loading silently wrong
    with pytest.raises(mp.ParamsFormatError):
        mp.from_gluon(g, exp, order='forward')


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


def test_legacy_flat_names_fall_back_to_forward_order():
    """Files this package's first exporter wrote: `arg:conv%d_weight` / `aux:batchnorm%d_running_mean`, ONE counter over all
    layers in forward order, no detection-block scopes.  They match the gluon name pattern but not a gluon CarNet's
    structure: from_gluon(order='auto') must map them by position (forward order) instead of rejecting them."""
    g = NetGraph(MICRO)
    P = _random_params(g, 3)
    legacy, ci, bi = {}, 0, 0
    for c in mp.gluon_conv_order(g, 'forward'):
        legacy['arg:conv%d_weight' % ci] = P[c.name + '.weight']
        if c.bn:
            legacy['arg:batchnorm%d_gamma' % bi] = P[c.name + '.gamma']
            legacy['arg:batchnorm%d_beta' % bi] = P[c.name + '.beta']
            legacy['aux:batchnorm%d_running_mean' % bi] = P[c.name + '.running_mean']
            legacy['aux:batchnorm%d_running_var' % bi] = P[c.name + '.running_var']
            bi += 1
        else:
            legacy['arg:conv%d_bias' % ci] = P[c.name + '.bias']
        ci += 1
    back = mp.from_gluon(g, legacy)
    assert set(back) == set(P)
    for k in P:
        np.testing.assert_array_equal(back[k], P[k], err_msg=k)
