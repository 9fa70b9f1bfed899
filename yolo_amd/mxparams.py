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
An exported symbol file (`arg:` / `aux:` prefixed names) lists arguments in forward (topological) order instead.
"""
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


def _suffix(name):
    for s in ('running_mean', 'running_var', 'moving_mean', 'moving_var', 'weight', 'gamma', 'beta', 'bias'):
        if name.endswith(s):
            return s.replace('moving', 'running')
    return None


def from_gluon(graph, loaded, order='auto'):
    """Map an OrderedDict read from a gluon `.params` file onto this package's names.  Parameters are matched by
    ORDER within their kind (the i-th conv weight of the file is the i-th conv in gluon order, and likewise the
    i-th gamma/beta/running_mean/running_var/bias) and every shape is checked, so the result does not depend on
    gluon's name counters."""
    names = list(loaded.keys())
    if order == 'auto':
        order = 'forward' if any(n.startswith(('arg:', 'aux:')) for n in names) else 'registration'
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


def to_gluon(graph, params, prefix='carnet0_'):
    """This package's parameters as an OrderedDict in collect_params() order with gluon-style names
    (<prefix>conv<i>_weight, <prefix>batchnorm<i>_gamma, ...; loading is by order, see from_gluon)."""
    out = OrderedDict()
    ci = bi = 0
    for c in gluon_conv_order(graph, 'registration'):
        get = lambda k: np.asarray(params[c.name + '.' + k].detach().cpu().numpy() if hasattr(params[c.name + '.' + k], 'detach')
                                   else params[c.name + '.' + k], np.float32)
        out['%sconv%d_weight' % (prefix, ci)] = get('weight')
        if c.bn:
            for s in ('gamma', 'beta', 'running_mean', 'running_var'):
                out['%sbatchnorm%d_%s' % (prefix, bi, s)] = get(s)
            bi += 1
        else:
            out['%sconv%d_bias' % (prefix, ci)] = get('bias')
        ci += 1
    return out
