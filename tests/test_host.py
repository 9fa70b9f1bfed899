"""CPU tests of the host-side mirror (no GPU, no compute calls): spec graph, grid descriptors, the
C-ABI library loads and exports every symbol include/yolo_amd.h declares, argument validation."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import graph as og, detect as od
from yolo_amd import spec as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graph_matches_oracle_graph():
    for sp in (og.spec_d53(), og.spec_car_v1(), og.spec_test_yaml(), og.spec_micro()):
        ng = S.NetGraph(sp)
        ref = og.conv_list(og.build_graph(sp))
        got = ng.convs()
        assert [(c.name, c.cin, c.cout, c.k, c.stride, c.bn) for c in got] == \
               [(c['name'], c['cin'], c['cout'], c['k'], c['stride'], c['bn']) for c in ref]
    ng = S.NetGraph(og.spec_d53())
    assert ng.flops(416, 416) == og.conv_flops(og.build_graph(og.spec_d53()), 416, 416)
    assert ng.steps() == [8, 16, 32]
    with pytest.raises(ValueError):
        S.NetGraph(dict(layers=[1, 2], channels=[8, 16], all_anchors=og.CAR_ANCHORS, slice_point=[1, 3, 5, 6, 10]))


def test_abi_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'yolo_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(yolo_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 18
    from yolo_amd import lib as L
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.yolo_version() == L.ABI_VERSION


def test_shipped_library_reads_no_environment_knobs():
    """The A/B and ablation switches of tools/ (YOLO_PIPE_PERSIST, YOLO_WW_*, ...) exist only in the lab build (`make lab`,
    -DYOLO_LAB): the product library must not change behaviour with the environment of whoever loads it, and exports no debug
    symbol."""
    from yolo_amd import lib as L
    blob = open(L.LIB_PATH, 'rb').read()
    names = set(re.findall(rb'YOLO_[A-Z0-9_]{2,}', blob))
    assert not names, names
    assert b'yolo_debug_read_stamps' not in blob


def test_abi_host_only_queries(lib):
    from yolo_amd import lib as L
    assert lib.yolo_padded_channels(90) % 128 == 0 and lib.yolo_padded_channels(90) >= 90
    # packed image: [chunk][tap][Cout_pad][64 B]
    assert lib.yolo_packed_weight_bytes(64, 32, 3, L.BF16) == 1 * 9 * lib.yolo_padded_channels(64) * 64
    assert lib.yolo_packed_weight_bytes(64, 32, 3, L.F32) == 2 * 9 * lib.yolo_padded_channels(64) * 64
    assert lib.yolo_packed_weight_bytes(64, 32, 5, L.BF16) < 0
    assert lib.yolo_nms_workspace_bytes(2, 10647, 24, 1, 400) == 2 * 10647 * 24 * 4 + lib.yolo_nms_select_workspace_bytes(2)
    assert lib.yolo_nms_select_workspace_bytes(2) > 0 and lib.yolo_nms_select_workspace_bytes(0) == -1
    # argument validation happens before any launch: NULL pointers / bad shapes are rejected on CPU too
    d = L.ConvDesc()
    assert lib.yolo_conv_fwd(C.byref(d), None) == -1
    assert lib.yolo_decode(None, None, 1, 30, None, None) == -1
    assert lib.yolo_nms_from_scores(None, None, 1, 1, 30, 1, 0.01, 0.45, 400, 100, None, None, None, None, None) == -1
    # kernel-name query is host-only
    d.x = d.w_packed = d.scale = d.bias = d.y = 1
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype = 32, 13, 13, 1024, 2048, 3, 1, L.BF16
    buf = C.create_string_buffer(256)
    assert lib.yolo_conv_kernel_name(C.byref(d), buf, 256) == 0
    assert b'conv_' in buf.value and b'bf16_t' in buf.value
    d.stride = 3
    assert lib.yolo_conv_kernel_name(C.byref(d), buf, 256) == -2


def test_grid_descriptor_matches_oracle():
    from yolo_amd.detect import make_grid
    sp = og.spec_d53()
    for size in ((416, 416), (320, 512), (608, 608)):
        steps = od.init_steps(sp['layers'], sp['all_anchors'])
        g, nbox = make_grid(sp['all_anchors'], size, steps)
        assert nbox == sum(od.init_area(size, steps)) * 3
        s, y, x, h, w = od.init_syxhw(size, steps, sp['all_anchors'])
        k = nbox // 2
        cell, a = divmod(k, 3)
        cum = 0
        for i in range(3):
            n = g.gh[i] * g.gw[i]
            if cell < cum + n:
                row, col = divmod(cell - cum, g.gw[i])
                assert (g.step[i], row * g.step[i], col * g.step[i]) == (s[0, cell, a, 0], y[0, cell, a, 0], x[0, cell, a, 0])
                assert abs(g.anchors_hw[(i * 3 + a) * 2] - h[0, cell, a, 0]) < 1e-7
                break
            cum += n


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under yolo_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'yolo_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'oracle.' not in src and 'oracle/' not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from yolo_amd import lib as L
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(L.YoloError):
        L.load()


def test_committed_bench_line_follows_the_contract():
    """profiles/r01_bench.json is the line bench.py printed on the GPU box: it must carry every field of the driver's
    contract, the roofline of the dominant kernel and the CPU baseline."""
    import json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r01_bench.json')
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'images/s' and d['scaling'] == 'weak' and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert d['data'] == 'synthetic' and d['dtype'] == 'bf16' and 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] in ('mfma', 'hbm') and r['unit'] in ('TFLOP/s', 'GB/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and (r['traffic'] is None or r['traffic'] > 0)
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    assert abs(d['value'] - d['n_gpus'] * d['config']['global_batch'] / d['n_gpus'] / (d['ms_per_step'] / 1e3)) / d['value'] < 0.01


def test_objects_refuse_a_foreign_current_device(monkeypatch):
    """One process per GPU: an object built for cuda:1 used while cuda:0 is current must be refused (its kernels would be
    launched on the wrong device); same device: no complaint."""
    import torch
    from yolo_amd import lib as L
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 0)
    L.require_current_device(torch.device('cuda:0'), 'x')
    L.require_current_device(torch.device('cuda'), 'x')
    with pytest.raises(L.YoloError, match='set_device'):
        L.require_current_device(torch.device('cuda:1'), 'this CarNet')


def test_conv_variant_eligibility_rules():
    """The GPU convolution tests are parametrised over the (shape, dtype, variant) pairs the library ACCEPTS
    (tests/util.py:eligible_pairs, a host-side query).  This test pins the other half on the CPU: which pairs are refused, and
    why -- a variant that starts refusing a shape it used to take (or the reverse) fails here instead of silently leaving the
    GPU parametrisation."""
    import test_gpu_conv as T
    from util import conv_variant
    sk = {30: (4, 3), 31: (2, 4), 32: (2, 3), 33: (2, 4), 35: (4, 4)}        # conv_sk.hip: (K groups, ring depth)
    n_ok = 0
    for case in T.PIPE_CASES:
        for dt in ('f32', 'bf16'):
            for a in T.PIPE_ALGOS:
                k = case[5]
                exp = True
                if a in (6, 7, 26, 27, 28) and k != 3:
                    exp = False                                               # 192-pixel, 128x128-wave and pixel-heavy tiles: 3x3 only
                if a == 26 and dt != 'bf16':
                    exp = False
                if a in (12, 19, 20, 21, 22, 23, 24, 25, 36, 37, 38, 39) and k != 1:
                    exp = False                                               # 1x1-only tiles / ring depths
                if a in sk:
                    kg, r = sk[a]
                    nch = case[1] * (4 if dt == 'f32' else 2) // 64
                    exp = k == 1 and nch % kg == 0 and nch // kg >= r - 1   # the K chunks divide over the wave groups
                got = conv_variant(case, dt, a) is not None
                assert got == exp, (case, dt, a, 'expected eligible' if exp else 'expected refused')
                n_ok += got
    assert n_ok == 283                                                        # (the GPU run exercises exactly these)
    # stride 2: every variant takes every test shape; streaming kernel: the wide variant (14) exists for stride 1, Cout >= 64
    assert all(conv_variant(c, dt, a) for c in T.S2_CASES for dt in ('f32', 'bf16') for a in T.S2_ALGOS)
    for c in T.STREAM_CASES:
        assert conv_variant(c, 'bf16', 13) is not None and conv_variant(c, 'f32', 13) is None
        assert (conv_variant(c, 'bf16', 14) is not None) == (c[6] == 1 and c[4] != 32), c
    # statistics epilogue: mode 2 (data-gradient sums) only on stride-1 shapes; the count is the GPU parametrisation's size
    n_stats = 0
    for c in T.STATS_CASES:
        for a in T.STATS_ALGOS:
            for m in (1, 2):
                ok = conv_variant(c, 'bf16', a, stats_mode=m) is not None
                assert not (ok and m == 2 and c[6] != 1), (c, a, m)
                n_stats += ok
    assert n_stats == 246


def test_launch_plan_round_trip_and_committed_file():
    """yolo_amd/plans.py: tuning states survive the JSON file (nested tuple keys, bools as 0 / 1), the md5 is over the contents, and
    the COMMITTED plan (profiles/plan.json, what bench.py launches by default) loads and holds choices for the three sections."""
    import tempfile
    from yolo_amd import plans
    st = {'algo': {(32, 13, 13, 1024, 512, 1, 1, 0, False, 1): 12, ('tail', 32, 52, 52, 128, 256, 1, True, 128, 0, 0, 1, 0, 0, 0): 1,
                   ('res', 32, 208, 208, 64): 1},
          'dgrad': {('s2', (64, 26, 26, 512), 512, 256, True): 6, ((64, 13, 13, 1024), 1024, 512, 3, False): 2},
          'wgrad': {(64, 13, 13, 512, 1024, 3, 1): 3}}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'p.json')
        plans.save(path, st, {'commit': 'x'})
        got, meta = plans.load(path)
        assert got == st and meta['md5'] == plans.md5(st) == plans.md5(got) and meta['commit'] == 'x'
        # a hand-edited plan is refused
        txt = open(path).read().replace('12', '13', 1)
        open(path, 'w').write(txt)
        with pytest.raises(ValueError):
            plans.load(path)
    assert plans.new_keys(plans.merge(st, {'algo': {(1, 2): 3}}), st) == 1 and plans.new_keys(st, st) == 0
    state, meta = plans.load(plans.DEFAULT)
    assert all(len(state[s]) > 20 for s in plans.SECTIONS), {s: len(state[s]) for s in plans.SECTIONS}
    assert meta['md5'] == plans.md5(state) and meta.get('commit')


def test_bench_host_probes():
    """bench.py's cpu_baseline placement inputs: the physical-core map covers only CPUs this process may use, one hardware thread per
    core; the CPU quota is None (unlimited) or a positive number of CPUs."""
    import bench
    topo = bench.host_topology()
    allowed = set(os.sched_getaffinity(0))
    cores = [c for v in topo.values() for c in v]
    assert cores and set(cores) <= allowed and len(cores) == len(set(cores))
    q = bench.cpu_quota()
    assert q is None or q > 0
