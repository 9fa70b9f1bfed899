"""MXNet `.params` files: reader / writer of the NDArray-list container that
`net.collect_params().save(path)` (car/YOLO.py:549) and `net.export` (yolo_gluon.py:257) write and
`collect_params().load(weight, ctx)` (yolo_gluon.py:190) reads, plus the mapping between gluon's parameter order
and this package's parameter names.  mxnet is not available here: the container is restated from its public layout
(mxnet/src/ndarray/ndarray.cc, NDArray::Save / NDArray::Load; little-endian):

    u64 0x112 (list magic)   u64 reserved (0)
    u64 n                    n x NDArray
    u64 n_names              n_names x { u64 length, bytes }

    NDArray (V2, magic 0xF993FAC9; V3 0xF993FACA has the same layout):
        u32 magic   i32 storage type (0 = dense; others unsupported here)
        u32 ndim    i64 dims[ndim]
        i32 dev_type   i32 dev_id                     (context the array was saved from; ignored)
        i32 type flag  (0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64)
        raw data, C order
    NDArray (V1, magic 0xF993FAC8): as V2 without the storage type.
    NDArray (legacy, no magic): u32 ndim, u32 dims[ndim], context, type flag, data.

Gluon parameter order (`collect_params()` is depth-first in child REGISTRATION order; basic_yolo.py:18-38 registers
`stages`, then `transitions`, `yolo_blocks`, `yolo_outputs`): stem, every stage (down-sampling conv, then its residual
blocks), the transitions, the detection blocks deep -> shallow (body convs, then tip), the outputs deep -> shallow.
Each `_conv2d` contributes weight, gamma, beta, running_mean, running_var; each YOLOOutput weight, bias.
`HybridBlock.export` (yolo_gluon.py:257) iterates the same `collect_params()` and only prefixes every name with
`arg:` / `aux:`, so an exported file is in registration order too.  The reference is Python-2 code (`exec "..."`), and
both writers go through a plain dict, so the order INSIDE a real file is hash order: parameters are therefore matched
by NAME (`gluon_param_names` below restates gluon's naming rules); position is only the fallback for files without
gluon names.

Gluon names (Block.__init__ -> _BlockScope.create, restated): a block created inside `with parent.name_scope()` is
called `<parent prefix><class name lower-cased><per-scope counter>_`; outside any scope the counter is the process-
wide NameManager's.  gluoncv's `_conv2d` and `DarknetBasicBlockV3` open NO scope of their own (their Conv2D /
BatchNorm take the counters of whatever scope is current), `YOLODetectionBlockV3` does.  BasicYOLONet builds the
backbone inside `with self.name_scope()` (basic_yolo.py:18-27) and calls YOLOPyrmaid OUTSIDE it (:32-37), and
YOLOOutput (basic_yolo.py:91-98) opens no scope, so a CarNet checkpoint holds

    carnet0_conv{0..51}_weight, carnet0_batchnorm{0..51}_{gamma,beta,running_mean,running_var}        backbone
    yolodetectionblockv3{k}_conv{0..5}_weight, ..._batchnorm{0..5}_*    detection block k (deep -> shallow; 5 = tip)
    conv{j}_weight, conv{j}_bias                    YOLOOutput k and, with batchnorm{t}_*, transition t, j in creation
                                                    order (output 0, output 1, transition 0, output 2, transition 1)

CarLPNet (car_and_LP/YOLO.py:47-60) adds yolodetectionblockv3{3..7}_* and one more biased top-level conv.  The
reader does not rely on the literal prefixes or counter values -- only on the structure (scope, layer kind, counter
order) -- and checks every shape.
"""
import re
import struct
from collections import OrderedDict

import numpy as np

LIST_MAGIC = 0x112
V1_MAGIC, V2_MAGIC, V3_MAGIC = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


class ParamsFormatError(ValueError):
    pass


class _Reader(object):
    def __init__(self, buf):
        self.b, self.o = memoryview(buf), 0

    def take(self, fmt):
        n = struct.calcsize(fmt)
        if self.o + n > len(self.b):
            raise ParamsFormatError('truncated file at byte %d' % self.o)
        v = struct.unpack_from(fmt, self.b, self.o)
        self.o += n
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        if self.o + n > len(self.b):
            raise ParamsFormatError('truncated array data at byte %d' % self.o)
        v = self.b[self.o:self.o + n]
        self.o += n
        return v


def _read_ndarray(r):
    magic = r.take('<I')
    if magic in (V2_MAGIC, V3_MAGIC):
        stype = r.take('<i')
        if stype != 0:
            raise ParamsFormatError('sparse storage type %d is not supported' % stype)
        ndim = r.take('<I')
        dims = [r.take('<q') for _ in range(ndim)]
    elif magic == V1_MAGIC:
        ndim = r.take('<I')
        dims = [r.take('<q') for _ in range(ndim)]
    else:                                   # legacy: the word just read is ndim, dims are u32
        ndim = magic
        if ndim > 32:
            raise ParamsFormatError('bad NDArray magic 0x%08x' % magic)
        dims = [r.take('<I') for _ in range(ndim)]
    if ndim == 0:
        return np.zeros((0,), np.float32)
    r.take('<ii')                           # context (dev_type, dev_id)
    flag = r.take('<i')
    if flag not in _DTYPES:
        raise ParamsFormatError('unknown type flag %d' % flag)
    dt = np.dtype(_DTYPES[flag]).newbyteorder('<')
    n = int(np.prod(dims, dtype=np.int64))
    return np.frombuffer(r.raw(n * dt.itemsize), dtype=dt).reshape(dims).copy()


def read_params(path):
    """-> OrderedDict name -> ndarray in file order (names '0','1',... when the file stores none)."""
    with open(path, 'rb') as f:
        r = _Reader(f.read())
    magic, _ = r.take('<QQ')
    if magic != LIST_MAGIC:
        raise ParamsFormatError('not an MXNet NDArray list (magic 0x%x)' % magic)
    n = r.take('<Q')
    arrays = [_read_ndarray(r) for _ in range(n)]
    nn = r.take('<Q')
    if nn not in (0, n):
        raise ParamsFormatError('%d names for %d arrays' % (nn, n))
    names = []
    for _ in range(nn):
        ln = r.take('<Q')
        names.append(bytes(r.raw(ln)).decode('utf-8'))
    if not names:
        names = [str(i) for i in range(n)]
    return OrderedDict(zip(names, arrays))


def write_params(path, params):
    """params: mapping name -> ndarray (written in iteration order, V2 arrays, cpu(0) context)."""
    out = [struct.pack('<QQQ', LIST_MAGIC, 0, len(params))]
    for a in params.values():
        a = np.ascontiguousarray(a)
        if a.dtype not in _FLAGS:
            raise ParamsFormatError('dtype %s has no MXNet type flag' % a.dtype)
        out.append(struct.pack('<IiI', V2_MAGIC, 0, a.ndim))
        out.append(struct.pack('<%dq' % a.ndim, *a.shape))
        out.append(struct.pack('<iii', 1, 0, _FLAGS[a.dtype]))
        out.append(a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes())
    out.append(struct.pack('<Q', len(params)))
    for n in params:
        b = n.encode('utf-8')
        out.append(struct.pack('<Q', len(b)) + b)
    with open(path, 'wb') as f:
        f.write(b''.join(out))


def gluon_conv_order(graph, order='registration'):
    """The net's convs in gluon order: 'registration' (collect_params().save) or 'forward' (exported symbol)."""
    convs = [graph.stem]
    for down, res in graph.stages:
        convs.append(down)
        for c1, c2 in res:
            convs += [c1, c2]
    heads = graph.heads                                   # deep -> shallow, as YOLOPyrmaid builds them
    lp = []
    for body, tip in getattr(graph, 'lp_blocks', []):
        lp += list(body) + [tip]
    if getattr(graph, 'lp_out', None) is not None:
        lp.append(graph.lp_out)
    if order == 'registration':
        convs += list(graph.transitions)
        for body, tip, out, _ in heads:
            convs += list(body) + [tip]
        convs += [out for _, _, out, _ in heads]
        convs += lp                                       # CarLPNet registers LP_branch after the base class's blocks
    elif order == 'forward':
        for i, (body, tip, out, _) in enumerate(heads):
            if i == len(heads) - 1:
                convs += lp                               # the LP branch runs before the finest detection block
            convs += list(body) + [tip, out]
            if i < len(graph.transitions):
                convs.append(graph.transitions[i])
    else:
        raise ValueError("order must be 'registration' or 'forward'")
    return convs


_SUFFIX = ('weight', 'gamma', 'beta', 'running_mean', 'running_var', 'bias')
_AUX = ('running_mean', 'running_var')
_NAME_RE = re.compile(r'^(?:(?:arg|aux):)?(.*?)(conv|batchnorm)(\d+)_(weight|bias|gamma|beta|running_mean|running_var|'
                      r'moving_mean|moving_var)$')
_BLOCK_RE = re.compile(r'^(.*)yolodetectionblockv3(\d+)_$')


def _suffix(name):
    for s in ('running_mean', 'running_var', 'moving_mean', 'moving_var', 'weight', 'gamma', 'beta', 'bias'):
        if name.endswith(s):
            return s.replace('moving', 'running')
    return None


def gluon_param_names(graph, prefix='carnet0_'):
    """OrderedDict `<our name>.<param>` -> gluon parameter name, in collect_params() (registration) order, for a net
    built as the first blocks of its process (fresh NameManager counters) -- see the module docstring."""
    out = OrderedDict()

    def unit(c, scope, ci, bi):
        out[c.name + '.weight'] = '%sconv%d_weight' % (scope, ci)
        if c.bn:
            for s in ('gamma', 'beta', 'running_mean', 'running_var'):
                out['%s.%s' % (c.name, s)] = '%sbatchnorm%d_%s' % (scope, bi, s)
        else:
            out[c.name + '.bias'] = '%sconv%d_bias' % (scope, ci)

    n = 0
    backbone = [graph.stem]
    for down, res in graph.stages:
        backbone.append(down)
        for c1, c2 in res:
            backbone += [c1, c2]
    for c in backbone:
        unit(c, prefix, n, n)
        n += 1
    # YOLOPyrmaid, outside every scope: per scale the output conv, the detection block, (i > 0) the transition
    top_conv, top_bn = {}, {}
    ci = bi = 0
    for i, (body, tip, outc, _) in enumerate(graph.heads):
        top_conv[outc.name] = ci; ci += 1
        if i > 0:
            t = graph.transitions[i - 1]
            top_conv[t.name], top_bn[t.name] = ci, bi
            ci += 1; bi += 1
    lp_blocks = list(getattr(graph, 'lp_blocks', []))
    lp_out = getattr(graph, 'lp_out', None)
    if lp_out is not None:
        top_conv[lp_out.name] = ci; ci += 1
    for t in graph.transitions:                                   # registration: transitions, yolo_blocks, yolo_outputs
        unit(t, '', top_conv[t.name], top_bn[t.name])
    for k, (body, tip, outc, _) in enumerate(graph.heads):
        for j, c in enumerate(list(body) + [tip]):
            unit(c, 'yolodetectionblockv3%d_' % k, j, j)
    for body, tip, outc, _ in graph.heads:
        unit(outc, '', top_conv[outc.name], None)
    for k, (body, tip) in enumerate(lp_blocks):                   # CarLPNet.LP_branch, registered last
        for j, c in enumerate(list(body) + [tip]):
            unit(c, 'yolodetectionblockv3%d_' % (len(graph.heads) + k), j, j)
    if lp_out is not None:
        unit(lp_out, '', top_conv[lp_out.name], None)
    return out


def _from_gluon_by_name(graph, loaded):
    """Structural name matching; returns None when the file's names are not gluon names."""
    scopes = {}
    for n, a in loaded.items():
        m = _NAME_RE.match(n)
        if m is None:
            return None
        scope, kind, idx, suffix = m.group(1), m.group(2), int(m.group(3)), m.group(4).replace('moving', 'running')
        layers = scopes.setdefault(scope, {'conv': {}, 'batchnorm': {}})[kind]
        ent = layers.setdefault(idx, {})
        if suffix in ent:
            raise ParamsFormatError('parameter %r appears twice' % n)
        ent[suffix] = np.asarray(a)

    def units(scope):
        """[(conv counter, {param: array})] of one scope: the k-th bias-less conv owns the k-th BatchNorm."""
        convs = sorted(scopes[scope]['conv'].items())
        bns = sorted(scopes[scope]['batchnorm'].items())
        plain = [(i, c) for i, c in convs if 'bias' not in c]
        if len(plain) != len(bns):
            raise ParamsFormatError('scope %r: %d convolutions without bias but %d BatchNorms' % (scope, len(plain), len(bns)))
        for (i, c), (_, b) in zip(plain, bns):
            c.update(b)
        return convs

    block_scopes, other = [], []
    for sc in scopes:
        m = _BLOCK_RE.match(sc)
        if m:
            block_scopes.append(((m.group(1), int(m.group(2))), sc))
        else:
            other.append(sc)
    block_scopes.sort()
    if not block_scopes and graph.heads:
        # flat conv%d / batchnorm%d names without a single detection-block scope: not a gluon CarNet's names (files this
        # package's first exporter wrote: one counter over all layers in forward order) -> positional mapping
        return None
    plain, biased = [], []
    # non-block scopes, the one with the most layers (the net's own: the backbone) first
    for sc in sorted(other, key=lambda sc_: (-len(scopes[sc_]['conv']), sc_)):
        for i, u in units(sc):
            (biased if 'bias' in u else plain).append(u)
    backbone = [graph.stem]
    for down, res in graph.stages:
        backbone.append(down)
        for c1, c2 in res:
            backbone += [c1, c2]
    lp_blocks = list(getattr(graph, 'lp_blocks', []))
    lp_out = getattr(graph, 'lp_out', None)
    want_plain = backbone + list(graph.transitions)
    want_biased = [h[2] for h in graph.heads] + ([lp_out] if lp_out is not None else [])
    want_blocks = [list(body) + [tip] for body, tip, _, _ in graph.heads] + [list(body) + [tip] for body, tip in lp_blocks]
    if len(plain) != len(want_plain) or len(biased) != len(want_biased) or len(block_scopes) != len(want_blocks):
        raise ParamsFormatError('file has %d conv+BN layers / %d biased convs / %d detection blocks outside the blocks; '
                                'the spec needs %d / %d / %d' % (len(plain), len(biased), len(block_scopes),
                                                                 len(want_plain), len(want_biased), len(want_blocks)))
    out = {}

    def put(c, u, where):
        need = ('weight', 'gamma', 'beta', 'running_mean', 'running_var') if c.bn else ('weight', 'bias')
        if sorted(u) != sorted(need):
            raise ParamsFormatError('%s (%s): file has %s, expected %s' % (c.name, where, sorted(u), sorted(need)))
        for k in need:
            shape = (c.cout, c.cin, c.k, c.k) if k == 'weight' else (c.cout,)
            if tuple(u[k].shape) != shape:
                raise ParamsFormatError('%s.%s (%s): expected shape %s, file has %s' % (c.name, k, where, shape, tuple(u[k].shape)))
            out['%s.%s' % (c.name, k)] = np.asarray(u[k], np.float32)

    for c, u in zip(want_plain, plain):
        put(c, u, 'conv+BN outside the detection blocks')
    for c, u in zip(want_biased, biased):
        put(c, u, 'biased output conv')
    for convs, (_, sc) in zip(want_blocks, block_scopes):
        us = units(sc)
        if len(us) != len(convs):
            raise ParamsFormatError('scope %r has %d layers, a detection block has %d' % (sc, len(us), len(convs)))
        for c, (_, u) in zip(convs, us):
            put(c, u, sc)
    return out


def from_gluon(graph, loaded, order='auto'):
    """Map an OrderedDict read from a gluon `.params` file onto this package's names.

    order='auto': by NAME when the file carries gluon parameter names (module docstring) -- the position inside the
    file is then irrelevant (real files are in Python-2 dict order); files without such names (e.g. '0', '1', ...) fall
    back to position in registration order.  order='registration' / 'forward' force the positional mapping: the i-th
    array of a kind is the i-th conv's in that order.  Every shape is checked."""
    if order == 'auto':
        byname = _from_gluon_by_name(graph, loaded)
        if byname is not None:
            return byname
        # flat gluon-looking names (no block scopes) are the legacy export of this package: forward order
        order = 'forward' if loaded and all(_NAME_RE.match(n) for n in loaded) else 'registration'
    names = list(loaded.keys())
    convs = gluon_conv_order(graph, order)
    by_kind = {s: [] for s in _SUFFIX}
    for n in names:
        s = _suffix(n)
        if s is None:
            raise ParamsFormatError('cannot classify parameter %r' % n)
        by_kind[s].append(loaded[n])
    out = {}
    idx = {s: 0 for s in _SUFFIX}

    def take(kind, shape, what):
        i = idx[kind]
        if i >= len(by_kind[kind]):
            raise ParamsFormatError('file has too few %s arrays (needed one for %s)' % (kind, what))
        a = by_kind[kind][i]
        idx[kind] += 1
        if tuple(a.shape) != tuple(shape):
            raise ParamsFormatError('%s: expected shape %s, file has %s' % (what, tuple(shape), tuple(a.shape)))
        return np.asarray(a, np.float32)

    for c in convs:
        out[c.name + '.weight'] = take('weight', (c.cout, c.cin, c.k, c.k), c.name + '.weight')
        if c.bn:
            for s in ('gamma', 'beta', 'running_mean', 'running_var'):
                out[c.name + '.' + s] = take(s, (c.cout,), c.name + '.' + s)
        else:
            out[c.name + '.bias'] = take('bias', (c.cout,), c.name + '.bias')
    for s in _SUFFIX:
        if idx[s] != len(by_kind[s]):
            raise ParamsFormatError('file has %d unused %s arrays' % (len(by_kind[s]) - idx[s], s))
    return out


def to_gluon(graph, params, prefix='carnet0_', export=False):
    """This package's parameters as an OrderedDict with gluon names (gluon_param_names) in collect_params() order.
    export=True: the `arg:` / `aux:` prefixes HybridBlock.export adds (running statistics are auxiliary states)."""
    out = OrderedDict()
    for ours, theirs in gluon_param_names(graph, prefix).items():
        v = params[ours]
        a = np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v, np.float32)
        if export:
            theirs = ('aux:' if ours.endswith(_AUX) else 'arg:') + theirs
        out[theirs] = a
    return out
