"""The SPLIT bf16 path (dtype 'bf16x3', include/yolo_amd.h YOLO_BF16X3; round 6): the arithmetic on which the north-star tolerance
-- decoded boxes within 1e-3 of the fp32 reference (car/YOLO.py:552-597) -- and the bf16 MFMA rate meet.

  * one convolution, every pipelined variant the library accepts for the type: against the path's own arithmetic restated on the
    CPU (operands as (hi, lo) bf16 pairs, three partial products, fp32 epilogue, the result a pair again: tests/util.py
    ref_conv_split) to a few units of the 16-bit storage, AND against the plain fp32 oracle at 2e-4;
  * the strided forms the net uses: a channel slice of a concat buffer as input, an up-sampled / sliced output, fp32 head logits;
  * the stem (direct fp32 convolution, split store);
  * whole nets: the D53 spec, the reference's car/v1 spec at its native 320x512, test.yaml and the micro spec (8 / 16-channel maps:
    padded planes) against the fp32 oracle's logits <= 1e-3 (tests/test_gpu_boxes.py holds the decoded boxes at 416 / 608).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import graph as og, forward as of
from util import run_conv, ref_conv, ref_conv_split, to_nhwc, from_nhwc, eligible_pairs, LDT, TDT, SPLIT
from yolo_amd import lib as L

pytestmark = pytest.mark.gpu

CASES = [
    (2, 128, 26, 26, 256, 3, 1, True),
    (4, 64, 52, 52, 256, 3, 1, True),      # several strips per row
    (8, 256, 13, 13, 512, 3, 1, True),     # tiles crossing image boundaries
    (3, 128, 19, 19, 256, 3, 1, False),    # 608-family odd map
    (1, 64, 13, 13, 96, 3, 1, False),      # fewer pixels than one tile, ragged Cout
    (1, 256, 13, 13, 128, 1, 1, False),
    (4, 256, 26, 26, 512, 1, 1, True),
    (2, 512, 13, 13, 88, 1, 1, False),
    (2, 64, 26, 26, 128, 3, 1, False),
    (2, 32, 40, 40, 64, 3, 1, True),       # one K-chunk per pass (3 chunks)
    (2, 32, 40, 40, 64, 3, 2, False),      # stride 2
    (3, 64, 26, 26, 128, 3, 2, False),
    (2, 128, 38, 38, 256, 3, 2, False),
    (2, 64, 20, 20, 32, 1, 1, False),      # 1x1, two chunks per pass
    (2, 96, 16, 16, 64, 1, 1, False),      # 1x1, three chunks per pass: a two-chunk phase would straddle the passes
    # planes padded to whole chunks (car/v1/spec.yaml's 16-channel maps, test.yaml's 8 / 16)
    (2, 16, 24, 24, 32, 3, 1, True),
    (2, 16, 20, 20, 32, 3, 2, False),
    (1, 8, 16, 16, 16, 3, 2, False),
    (2, 32, 16, 16, 16, 1, 1, False),      # padded OUTPUT planes
    (2, 16, 12, 12, 8, 1, 1, False),
    (2, 48, 13, 13, 64, 1, 1, False),
    (2, 8, 16, 16, 16, 3, 1, True),        # padded input, output and residual
    # prime widths / tiny maps: no strip of the pipelined kernels fits -- the generic kernel (algo 1; algo 0 falls back to it)
    (2, 64, 9, 31, 128, 3, 1, True),
    (3, 32, 21, 37, 64, 3, 1, False),
    (1, 128, 30, 47, 64, 3, 1, False),
    (2, 64, 7, 43, 64, 3, 2, False),
    (2, 40, 5, 11, 24, 1, 1, False),
    (1, 72, 1, 1, 48, 3, 1, False),
]
ALGOS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28, 36, 37, 38, 39, 40, 41, 42]


def _mk(case, seed):
    N, Cin, H, W, Cout, k, stride, res = case
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32) * 0.1
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    r = rng.standard_normal((N, Cout, Ho, Wo)).astype(np.float32) if res else None
    return x, w, scale, bias, r


def _check(y, sim, ref, dtype='bf16x3'):
    assert not np.isnan(y).any()
    # against the restated arithmetic: fp32 accumulation order + the 16-bit (f16x3: 22-bit) storage, one unit either way
    np.testing.assert_allclose(y, sim, rtol=3e-5, atol=3e-5)
    # against plain fp32: the dropped lo x lo term and the operands' storage (f16x3: fp32's own accumulation-order noise)
    tol = 2e-4 if dtype == 'bf16x3' else 3e-5
    np.testing.assert_allclose(y, ref, rtol=tol, atol=tol)


def _sim(dtype, *a, **kw):
    return ref_conv_split(*a, rdt=TDT[dtype], **kw)


@pytest.mark.parametrize('case,dtype,algo', eligible_pairs(CASES, list(SPLIT), ALGOS))
def test_split_conv_variants(lib, cuda, case, dtype, algo):
    x, w, scale, bias, r = _mk(case, 4)
    y = run_conv(lib, cuda, x, w, scale, bias, case[6], 0.1, dtype, residual=r, algo=algo, expect_rc=0)
    _check(y, _sim(dtype, x, w, scale, bias, case[6], 0.1, residual=r), ref_conv(x, w, scale, bias, case[6], 0.1, residual=r), dtype)


@pytest.mark.parametrize('dtype', SPLIT)
@pytest.mark.parametrize('algo', [42, 17, 1])
def test_split_stride2_full_size_repeated(lib, cuda, algo, dtype):
    """The D53 104x104 -> 52x52 down-sampling layer at its real size, five launches each: the 4-wave 64-cout stride-2 tile (algo 42)
    stages ten input DMAs per thread and chunk -- its first cut issued one of them in a chunk's last phase, where the counted wait
    still lets it fly, and computed on a stale unit now and then (caught by the whole-net test, not by the small conv cases)."""
    case = (2, 128, 104, 104, 256, 3, 2, False)
    x, w, scale, bias, r = _mk(case, 12)
    sim, ref = _sim(dtype, x, w, scale, bias, 2, 0.1), ref_conv(x, w, scale, bias, 2, 0.1)
    for _ in range(5):
        _check(run_conv(lib, cuda, x, w, scale, bias, 2, 0.1, dtype, algo=algo, expect_rc=0), sim, ref, dtype)


@pytest.mark.parametrize('dtype', SPLIT)
@pytest.mark.parametrize('case', CASES)
def test_split_conv_auto(lib, cuda, case, dtype):
    """algo 0 (the library's heuristic) takes every one of these shapes."""
    x, w, scale, bias, r = _mk(case, 9)
    y = run_conv(lib, cuda, x, w, scale, bias, case[6], 0.1, dtype, residual=r, algo=0, expect_rc=0)
    _check(y, _sim(dtype, x, w, scale, bias, case[6], 0.1, residual=r), ref_conv(x, w, scale, bias, case[6], 0.1, residual=r), dtype)


def test_split_refusals(lib, cuda):
    """What the type does not cover fails loudly: channel counts that are not multiples of 8, the streaming / split-K kernels,
    statistics and fused tails, the training entries."""
    x, w, scale, bias, _ = _mk((1, 64, 13, 13, 20, 1, 1, False), 1)
    run_conv(lib, cuda, x, w, scale, bias, 1, 0.1, 'bf16x3', expect_rc=L.EUNSUPPORTED)      # Cout % 8 (16-byte store pieces)
    assert lib.yolo_packed_weight_bytes(32, 12, 3, L.BF16X3) == L.EUNSUPPORTED
    x, w, scale, bias, _ = _mk((1, 64, 13, 13, 64, 1, 1, False), 1)
    for algo in (13, 30, 31, 26):
        run_conv(lib, cuda, x, w, scale, bias, 1, 0.1, 'bf16x3', algo=algo, expect_rc=L.EUNSUPPORTED)
    assert lib.yolo_pack_batch_blocks(64, 64, 3, L.BF16X3) == L.EUNSUPPORTED
    assert lib.yolo_conv_wgrad_workspace_bytes(64, 64, 3, L.BF16X3) == L.EINVAL
    assert lib.yolo_packed_weight_bytes(64, 64, 3, 7) == L.EINVAL and lib.yolo_pack_batch_blocks(64, 64, 3, 7) == L.EINVAL


def test_split_out_f32_and_views(lib, cuda):
    """The forms CarNet uses around the detection blocks: fp32 head logits written with strides into a merged buffer (YOLOOutput,
    basic_yolo.py:98-105), the input a channel slice of a wider split buffer and the output up-sampled 2x into one half of a
    concat buffer (car/utils.py:91-93)."""
    st = torch.cuda.current_stream().cuda_stream
    dt = L.BF16X3
    # --- head logits
    case = (2, 64, 13, 13, 90, 1, 1, False)
    x, w, scale, bias, _ = _mk(case, 3)
    scale[:] = 1.0
    y = run_conv(lib, cuda, x, w, scale, bias, 1, 1.0, 'bf16x3', out_f32=True)
    np.testing.assert_allclose(y, ref_conv_split(x, w, scale, bias, 1, 1.0, out_f32=True), rtol=3e-5, atol=3e-5)
    np.testing.assert_allclose(y, ref_conv(x, w, scale, bias, 1, 1.0), rtol=2e-4, atol=2e-4)
    # --- 1x1 transition: input = channels [32, 96) of a 128-channel split buffer, output up-sampled into channels [0, 32) of a
    #     96-channel split buffer whose other channels must stay untouched
    N, H, W, Ct, c0, Cin, Cout, Co_t = 2, 10, 12, 128, 32, 64, 32, 96
    rng = np.random.default_rng(8)
    xb = rng.standard_normal((N, Ct, H, W)).astype(np.float32)
    wgt = (rng.standard_normal((Cout, Cin, 1, 1)) / 8).astype(np.float32)
    sc, bi = rng.uniform(0.5, 1.5, Cout).astype(np.float32), (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    xd = to_nhwc(xb, 'bf16x3', cuda)                                        # (N, H, W, 2, Ct)
    xv = xd[..., c0:c0 + Cin]
    cat = torch.full((N, 2 * H, 2 * W, 2, Co_t), 7.0, dtype=torch.bfloat16, device=cuda)
    ov = cat[..., :Cout]
    wp = torch.empty(lib.yolo_packed_weight_bytes(Cout, Cin, 1, dt), dtype=torch.uint8, device=cuda)
    L.check(lib.yolo_pack_conv_weights(torch.from_numpy(wgt).to(cuda).data_ptr(), wp.data_ptr(), Cout, Cin, 1, dt, st), 'pack')
    cp = lib.yolo_padded_channels(Cout)
    scd = torch.zeros(cp, device=cuda); scd[:Cout] = torch.from_numpy(sc).to(cuda)
    bid = torch.zeros(cp, device=cuda); bid[:Cout] = torch.from_numpy(bi).to(cuda)
    d = L.ConvDesc()
    d.x, d.w_packed, d.scale, d.bias, d.y = xv.data_ptr(), wp.data_ptr(), scd.data_ptr(), bid.data_ptr(), ov.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope = N, H, W, Cin, Cout, 1, 1, dt, 0.1
    d.x_pixel_stride, d.x_lo_offset = xv.stride(2), xv.stride(3)
    d.y_pixel_stride, d.y_batch_stride, d.y_lo_offset, d.upsample2x = ov.stride(2), ov.stride(0), ov.stride(3), 1
    L.check(lib.yolo_conv_fwd(C.byref(d), st), 'conv')
    torch.cuda.synchronize()
    want = ref_conv_split(xb[:, c0:c0 + Cin], wgt, sc, bi, 1, 0.1)
    want = np.repeat(np.repeat(want, 2, axis=2), 2, axis=3)                # gluoncv _upsample: repeat on W then H
    got = from_nhwc(cat)
    np.testing.assert_allclose(got[:, :Cout], want, rtol=3e-5, atol=3e-5)
    assert bool((cat[..., Cout:] == 7.0).all())                            # the route half of both planes untouched
    # descriptors that do not describe a split tensor are refused
    d.x_lo_offset = Cin - 8
    assert lib.yolo_conv_fwd(C.byref(d), st) == L.EINVAL


@pytest.mark.parametrize('dtype', SPLIT)
def test_split_stem(lib, cuda, dtype):
    """yolo_stem_conv_fwd, YOLO_BF16X3 / YOLO_F16X3: direct fp32 convolution of the NCHW image, output stored split."""
    st = torch.cuda.current_stream().cuda_stream
    for (N, H, W, Cout) in [(2, 33, 47, 32), (1, 16, 16, 64), (3, 8, 20, 8)]:
        rng = np.random.default_rng(N + Cout)
        x = rng.random((N, 3, H, W), dtype=np.float32)
        w = (rng.standard_normal((Cout, 3, 3, 3)) / 5).astype(np.float32)
        sc, bi = rng.uniform(0.5, 1.5, Cout).astype(np.float32), (rng.standard_normal(Cout) * 0.1).astype(np.float32)
        Cp = -(-Cout // 32) * 32
        y = torch.zeros((N, H, W, 2, Cp), dtype=TDT[dtype], device=cuda)
        t = lambda a: torch.from_numpy(a).to(cuda)
        xd, wd, sd, bd = t(x), t(w), t(sc), t(bi)
        L.check(lib.yolo_stem_conv_fwd(xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), bd.data_ptr(), y.data_ptr(), N, H, W, 3, Cout,
                                       LDT[dtype], 0.1, st), 'stem')
        torch.cuda.synchronize()
        ref = ref_conv(x, w, sc, bi, 1, 0.1)
        np.testing.assert_allclose(from_nhwc(y)[:, :Cout], ref, rtol=2e-5, atol=2e-5)   # exact fp32 products; the storage's 2^-17
        assert bool((y[..., Cout:] == 0).all())                                      # the planes' pad channels stay as the caller zeroed them
    assert lib.yolo_stem_conv_fwd(xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), bd.data_ptr(), y.data_ptr(), 1, 8, 8, 3, 12,
                                  LDT[dtype], 0.1, st) == L.EUNSUPPORTED


@pytest.mark.parametrize('dtype', SPLIT)
@pytest.mark.parametrize('which,size,B', [('micro', (64, 64), 3), ('test_yaml', (192, 256), 2), ('car_v1', (320, 512), 2)])
def test_split_net_narrow_specs(cuda, which, size, B, dtype):
    """Nets whose early maps have 8 / 16 channels -- the reference's own car/v1/spec.yaml at its native 320x512, yolo_modules/
    test.yaml at 192x256 (basic_yolo.py:129-133), the micro spec: split planes are padded to whole 32-channel chunks.  Logits within
    1e-3 of the fp32 oracle."""
    from yolo_amd.net import CarNet
    spec = {'micro': og.spec_micro, 'test_yaml': og.spec_test_yaml, 'car_v1': og.spec_car_v1}[which]()
    g = og.build_graph(spec)
    P = og.init_params(g, seed=0, bn='random')
    x = np.random.default_rng(3).random((B, 3) + size, dtype=np.float32)
    ref = [r.numpy() for r in of.forward_torch(g, P, x)]
    net = CarNet(spec, dtype=dtype, device=cuda).load_params(P)
    outs = net(torch.from_numpy(x).to(cuda))
    for o, r in zip(outs, ref):
        np.testing.assert_allclose(o.cpu().numpy(), r, rtol=0, atol=1e-3 if dtype == 'bf16x3' else 5e-4)      # (f16x3: fp32's own noise on logits of |t| ~ 20)


@pytest.mark.parametrize('dtype', SPLIT)
@pytest.mark.parametrize('tune', ['auto', 'measure'])
def test_split_d53_logits_vs_fp32_oracle(cuda, tune, dtype):
    """D53 spec at 416x416, random BN: every head logit within 1e-3 of the fp32 oracle (observed ~3e-4 at |logit| up to 19), taps
    along the way within the storage's few units; the same batch twice is bit-identical (no atomics on the path)."""
    from yolo_amd.net import CarNet
    spec = og.spec_d53()
    g = og.build_graph(spec)
    P = og.init_params(g, seed=0, bn='random')
    x = np.random.default_rng(2).random((2, 3, 416, 416), dtype=np.float32)
    taps = {}
    ref = [r.numpy() for r in of.forward_torch(g, P, x, taps=taps)]
    net = CarNet(spec, dtype=dtype, device=cuda, tune=tune).load_params(P)
    xt = torch.from_numpy(x).to(cuda)
    outs = [o.clone() for o in net(xt)]
    assert all(kind in ('conv', 'stem') for kind, _, _ in net._last_plan.ops), 'a single-plane kernel in the split plan'
    stem = net.activation_nchw('stem').cpu().numpy()
    np.testing.assert_allclose(stem, taps['stem'].numpy(), rtol=2e-5, atol=2e-5)
    for i in range(len(g['stages'])):
        last = g['stages'][i]['res'][-1][1]['name'] if g['stages'][i]['res'] else g['stages'][i]['down']['name']
        got, want = net.activation_nchw(last).cpu().numpy(), taps['stages.%d' % i].numpy()
        assert np.abs(got - want).max() <= 2e-4 * (1 + np.abs(want).max()), (i, np.abs(got - want).max())
    for o, r in zip(outs, ref):
        np.testing.assert_allclose(o.cpu().numpy(), r, rtol=0, atol=1e-3 if dtype == 'bf16x3' else 2e-4)      # (f16x3: the fp32 path's own distance)
    again = net(xt)
    assert all(bool((a == b).all()) for a, b in zip(again, outs))
