"""Unit parity of the backward building blocks against torch-CPU autograd: weight gradient (MFMA f32),
data gradient (forward kernel on the flipped weight image, dilation for stride 2), train-mode BatchNorm
forward/backward, up-sample/concat backward."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from yolo_amd import lib as L
from util import to_nhwc, from_nhwc

pytestmark = pytest.mark.gpu

# (N, Cin, H, W, Cout, k, stride)
SHAPES = [(2, 64, 8, 12, 128, 3, 1), (2, 128, 8, 12, 64, 1, 1), (3, 32, 13, 13, 64, 3, 1), (2, 16, 16, 24, 32, 3, 2),
          (2, 64, 4, 6, 30, 1, 1), (1, 8, 20, 20, 16, 3, 1), (2, 96, 2, 3, 128, 3, 1), (2, 32, 26, 26, 64, 3, 2)]


def _ref(case, seed):
    N, Cin, H, W, Cout, k, s = case
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((N, Cin, H, W)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=k // 2)
    dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(dy)
    return x.detach().numpy(), w.detach().numpy(), dy.numpy(), x.grad.numpy(), w.grad.numpy()


@pytest.mark.parametrize('case', SHAPES)
def test_wgrad(lib, cuda, case):
    N, Cin, H, W, Cout, k, s = case
    x, w, dy, dx_ref, dw_ref = _ref(case, 1)
    xd, dyd = to_nhwc(x, 'f32', cuda), to_nhwc(dy, 'f32', cuda)
    dw = torch.zeros((Cout, Cin, k, k), device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.yolo_conv_wgrad(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, k, s, 0, L.F32, None, st) == 0
    np.testing.assert_allclose(dw.cpu().numpy(), dw_ref, rtol=1e-3, atol=1e-3 * np.abs(dw_ref).max())


@pytest.mark.parametrize('case', SHAPES)
def test_dgrad(lib, cuda, case):
    N, Cin, H, W, Cout, k, s = case
    if Cout % 4:
        pytest.skip('dy channel count must be a multiple of 4 (the trainer pads the head logits)')
    x, w, dy, dx_ref, dw_ref = _ref(case, 2)
    st = torch.cuda.current_stream().cuda_stream
    wd = torch.empty(lib.yolo_packed_weight_bytes(Cin, Cout, k, L.F32), dtype=torch.uint8, device=cuda)
    assert lib.yolo_pack_conv_weights_dgrad(torch.from_numpy(w).to(cuda).data_ptr(), wd.data_ptr(), Cout, Cin, k, L.F32, st) == 0
    dyd = to_nhwc(dy, 'f32', cuda)
    if s == 2:
        dil = torch.empty((N, H, W, Cout), device=cuda)
        assert lib.yolo_dilate2x(dyd.data_ptr(), dil.data_ptr(), N, H, W, dy.shape[2], dy.shape[3], Cout, L.F32, st) == 0
        dyd = dil
    cp = lib.yolo_padded_channels(Cin)
    ones = torch.ones(cp, device=cuda); zeros = torch.zeros(cp, device=cuda)
    out = torch.full((N, H, W, Cin), float('nan'), device=cuda)
    d = L.ConvDesc()
    d.x, d.w_packed, d.scale, d.bias, d.y = dyd.data_ptr(), wd.data_ptr(), ones.data_ptr(), zeros.data_ptr(), out.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope = N, H, W, Cout, Cin, k, 1, L.F32, 1.0
    assert lib.yolo_conv_fwd(C.byref(d), st) == 0
    np.testing.assert_allclose(from_nhwc(out), dx_ref, rtol=1e-3, atol=1e-3 * np.abs(dx_ref).max())
    # accumulate into an existing gradient (residual = y, in place)
    d.residual = out.data_ptr()
    assert lib.yolo_conv_fwd(C.byref(d), st) == 0
    np.testing.assert_allclose(from_nhwc(out), 2 * dx_ref, rtol=1e-3, atol=2e-3 * np.abs(dx_ref).max())


@pytest.mark.parametrize('shape', [(2, 8, 12, 128), (3, 13, 13, 64), (1, 4, 6, 32), (4, 32, 48, 16), (2, 5, 7, 2048 + 64),
                                   # D53 shapes: the widest map (one 256-channel group, many pixel ranges) and the widest
                                   # layer over the smallest map (eight channel groups)
                                   (2, 208, 208, 64), (8, 13, 13, 2048), (8, 52, 52, 256)])
@pytest.mark.parametrize('with_res', [False, True])
def test_bn_train_fwd_bwd(lib, cuda, shape, with_res):
    N, H, W, Cc = shape
    rng = np.random.default_rng(3)
    y = torch.from_numpy((2 * rng.standard_normal((N, Cc, H, W)) + 0.5).astype(np.float32)).requires_grad_(True)
    gamma = torch.from_numpy(rng.uniform(.5, 1.5, Cc).astype(np.float32)).requires_grad_(True)
    beta = torch.from_numpy((.1 * rng.standard_normal(Cc)).astype(np.float32)).requires_grad_(True)
    res = torch.from_numpy(rng.standard_normal((N, Cc, H, W)).astype(np.float32)) if with_res else None
    mean = y.mean(dim=(0, 2, 3)); var = y.var(dim=(0, 2, 3), unbiased=False)
    a = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    z = F.leaky_relu(a, 0.1)
    if with_res:
        z = z + res
    dz = torch.from_numpy(rng.standard_normal((N, Cc, H, W)).astype(np.float32))
    z.backward(dz)
    st = torch.cuda.current_stream().cuda_stream
    dev = lambda t: t.detach().to(cuda).contiguous()
    yd, resd, dzd = to_nhwc(y.detach().numpy(), 'f32', cuda), (to_nhwc(res.numpy(), 'f32', cuda) if with_res else None), to_nhwc(dz.numpy(), 'f32', cuda)
    g_, b_ = dev(gamma), dev(beta)
    zd = torch.empty_like(yd); m_ = torch.empty(Cc, device=cuda); is_ = torch.empty(Cc, device=cuda)
    rm = torch.zeros(Cc, device=cuda); rv = torch.ones(Cc, device=cuda)
    ws = torch.zeros(3 * Cc, dtype=torch.float64, device=cuda)
    npix = N * H * W
    assert lib.yolo_bn_train_fwd(yd.data_ptr(), g_.data_ptr(), b_.data_ptr(), resd.data_ptr() if with_res else None,
                                 zd.data_ptr(), m_.data_ptr(), is_.data_ptr(), rm.data_ptr(), rv.data_ptr(), ws.data_ptr(),
                                 npix, Cc, 1e-5, 0.9, 0.1, L.F32, st) == 0
    np.testing.assert_allclose(from_nhwc(zd), z.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m_.cpu().numpy(), mean.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), 0.9 + 0.1 * var.detach().numpy(), rtol=1e-5)
    dyd = torch.empty_like(yd); dg = torch.empty(Cc, device=cuda); db = torch.empty(Cc, device=cuda)
    assert lib.yolo_bn_train_bwd(dzd.data_ptr(), yd.data_ptr(), m_.data_ptr(), is_.data_ptr(), g_.data_ptr(), b_.data_ptr(),
                                 dyd.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), npix, Cc, 0.1, L.F32, st) == 0
    np.testing.assert_allclose(from_nhwc(dyd), y.grad.numpy(), rtol=1e-3, atol=1e-4 * np.abs(y.grad.numpy()).max())
    np.testing.assert_allclose(dg.cpu().numpy(), gamma.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), beta.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_upcat_bwd(lib, cuda):
    rng = np.random.default_rng(4)
    up = torch.from_numpy(rng.standard_normal((2, 8, 4, 6)).astype(np.float32)).requires_grad_(True)
    route = torch.from_numpy(rng.standard_normal((2, 12, 8, 12)).astype(np.float32)).requires_grad_(True)
    cat = torch.cat([up.repeat_interleave(2, -1).repeat_interleave(2, -2), route], dim=1)
    dcat = torch.from_numpy(rng.standard_normal(tuple(cat.shape)).astype(np.float32))
    cat.backward(dcat)
    st = torch.cuda.current_stream().cuda_stream
    dc = to_nhwc(dcat.numpy(), 'f32', cuda)
    dup = torch.empty((2, 4, 6, 8), device=cuda); dr = torch.ones((2, 8, 12, 12), device=cuda)
    assert lib.yolo_upsample2x_concat_bwd(dc.data_ptr(), dup.data_ptr(), dr.data_ptr(), 2, 8, 12, 8, 12, 0, 1, L.F32, st) == 0
    np.testing.assert_allclose(from_nhwc(dup), up.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(from_nhwc(dr), route.grad.numpy() + 1.0, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('case', [(2, 64, 8, 12, 128, 3, 1), (2, 128, 13, 13, 64, 1, 1), (3, 32, 13, 13, 64, 3, 1),
                                  (2, 16, 16, 24, 32, 3, 2), (2, 64, 4, 6, 96, 1, 1), (1, 8, 20, 20, 16, 3, 1),
                                  (4, 256, 26, 26, 192, 3, 1), (2, 32, 26, 26, 64, 3, 2),
                                  # strip kernel (3x3, Cin <= 64): several strips / row slices, ragged widths, channel tails
                                  (1, 8, 70, 45, 32, 3, 1), (2, 40, 37, 66, 72, 3, 1), (1, 64, 50, 33, 128, 3, 2),
                                  (1, 24, 41, 71, 40, 3, 2), (40, 32, 9, 9, 64, 3, 1),
                                  # row-group kernel (3x3 stride 1, Cin > 64): every (TH, K-steps) variant, ragged
                                  # heights (H % TH != 0), channel tails, several images and row slices
                                  (3, 128, 13, 13, 96, 3, 1), (2, 72, 17, 19, 200, 3, 1), (2, 128, 11, 26, 64, 3, 1),
                                  (1, 80, 9, 38, 72, 3, 1), (1, 96, 7, 32, 64, 3, 1), (1, 128, 5, 40, 64, 3, 1),
                                  (2, 128, 6, 16, 64, 3, 1), (1, 72, 3, 64, 64, 3, 1), (1, 72, 4, 100, 64, 3, 1),
                                  (70, 128, 13, 13, 128, 3, 1),
                                  # the D53 layer shapes of BASELINE configs[2] themselves (416x416; batch 8): the strip kernel
                                  # at 416^2 / 208^2, the split-pixel-range atomics of the per-tap kernel, the row groups at
                                  # 52^2 / 26^2 / 13^2, the widest 1x1 and 3x3 layers
                                  (8, 8, 416, 416, 32, 3, 1), (8, 32, 416, 416, 64, 3, 2), (8, 64, 208, 208, 32, 1, 1),
                                  (8, 32, 208, 208, 64, 3, 1), (8, 64, 208, 208, 128, 3, 2), (8, 128, 52, 52, 256, 3, 1),
                                  (8, 256, 52, 52, 128, 1, 1), (8, 256, 26, 26, 512, 3, 1), (8, 512, 13, 13, 1024, 3, 1),
                                  (8, 1024, 13, 13, 2048, 3, 1), (8, 2048, 13, 13, 1024, 1, 1), (8, 512, 26, 26, 1024, 3, 2)])
def test_wgrad_bf16_transposing_reads(lib, cuda, case):
    """bf16 weight gradient (MFMA 32x32x16 fed by ds_read_b64_tr_b16): exact fp32 accumulation of the
    bf16-rounded operands, so it must match torch on the rounded inputs to fp32 noise."""
    N, Cin, H, W, Cout, k, s = case
    x, w, dy, dx_ref, dw_ref = _ref(case, 3)
    rb = lambda a: torch.from_numpy(a).to(torch.bfloat16).float()
    xr = rb(x).requires_grad_(False); wt = torch.from_numpy(w).requires_grad_(True)
    y = F.conv2d(xr, wt, None, stride=s, padding=k // 2)
    y.backward(rb(dy))
    xd, dyd = to_nhwc(x, 'bf16', cuda), to_nhwc(dy, 'bf16', cuda)
    dw = torch.ones((Cout, Cin, k, k), device=cuda)               # accumulates into the existing gradient
    ws = torch.zeros(lib.yolo_conv_wgrad_workspace_bytes(Cin, Cout, k, L.BF16), dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.yolo_conv_wgrad(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, k, s, 0, L.BF16,
                               ws.data_ptr(), st) == 0
    ref = wt.grad.numpy()
    np.testing.assert_allclose(dw.cpu().numpy() - 1.0, ref, rtol=1e-3, atol=2e-3 * np.abs(ref).max())


@pytest.mark.parametrize('algo', [5, 6])
@pytest.mark.parametrize('case', [(2, 128, 8, 12, 128), (3, 256, 13, 13, 128), (1, 128, 5, 13, 256), (4, 384, 9, 7, 256),
                                  (70, 128, 13, 13, 128), (8, 256, 52, 52, 128), (8, 512, 26, 26, 256), (8, 1024, 13, 13, 512),
                                  (8, 2048, 13, 13, 1024), (2, 768, 38, 38, 256)])
def test_wgrad_bf16_1x1_gemm(lib, cuda, case, algo):
    """wgrad_gemm_kernel (1x1 layers, Cin and Cout multiples of 128): LDS-DMA ring + transposing fragment reads, K-slices
    ending inside a phase (zero page), partial sums added straight into the gradient; both tile variants."""
    N, Cin, H, W, Cout = case
    x, w, dy, dx_ref, dw_ref = _ref((N, Cin, H, W, Cout, 1, 1), 13)
    rb = lambda a: torch.from_numpy(a).to(torch.bfloat16).float()
    wt = torch.from_numpy(w).requires_grad_(True)
    F.conv2d(rb(x), wt, None).backward(rb(dy))
    xd, dyd = to_nhwc(x, 'bf16', cuda), to_nhwc(dy, 'bf16', cuda)
    dw = torch.ones((Cout, Cin, 1, 1), device=cuda)
    ws = torch.zeros(lib.yolo_conv_wgrad_workspace_bytes(Cin, Cout, 1, L.BF16), dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.yolo_conv_wgrad_algo(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, 1, 1, 0, L.BF16, ws.data_ptr(), algo, st)
    if algo == 6 and Cout % 256:
        assert rc == L.EUNSUPPORTED
        return
    assert rc == 0
    ref = wt.grad.numpy()
    np.testing.assert_allclose(dw.cpu().numpy() - 1.0, ref, rtol=1e-3, atol=2e-3 * np.abs(ref).max())
    assert lib.yolo_conv_wgrad_algo(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, 3, 1, 0, L.BF16,
                                    ws.data_ptr(), algo, st) == L.EUNSUPPORTED


@pytest.mark.parametrize('algo', [2, 3, 4])
@pytest.mark.parametrize('case', [(2, 64, 8, 12, 128), (3, 128, 13, 13, 64), (2, 64, 17, 19, 192), (1, 64, 5, 4, 64),
                                  (2, 128, 11, 26, 64), (1, 64, 9, 38, 128), (5, 64, 7, 33, 64), (1, 64, 2, 70, 64),
                                  (70, 64, 13, 13, 128), (3, 192, 30, 52, 64),
                                  # D53 shapes of BASELINE configs[2] (416x416; batch 8) and the 608x608 family's widths
                                  (8, 64, 104, 104, 128), (8, 128, 52, 52, 256), (8, 256, 26, 26, 512),
                                  (8, 512, 13, 13, 1024), (4, 256, 38, 38, 512), (4, 512, 19, 19, 1024),
                                  (2, 128, 76, 76, 256)])
def test_wgrad_bf16_row_walk(lib, cuda, case, algo):
    """wgrad_walk_kernel (3x3 stride 1, Cin and Cout multiples of 64): LDS-DMA ring, x fragments of the three kernel
    rows held in registers, walkers over the stacked padded rows -- one 16-column walker (algo 2) and four 4-column
    walkers (algo 3) per block; ragged widths, walker ranges that cross image boundaries, several row slices.  Exact
    fp32 accumulation of the bf16-rounded operands, as the other weight-gradient kernels."""
    N, Cin, H, W, Cout = case
    full = (N, Cin, H, W, Cout, 3, 1)
    x, w, dy, dx_ref, dw_ref = _ref(full, 11)
    rb = lambda a: torch.from_numpy(a).to(torch.bfloat16).float()
    wt = torch.from_numpy(w).requires_grad_(True)
    F.conv2d(rb(x), wt, None, stride=1, padding=1).backward(rb(dy))
    xd, dyd = to_nhwc(x, 'bf16', cuda), to_nhwc(dy, 'bf16', cuda)
    dw = torch.ones((Cout, Cin, 3, 3), device=cuda)
    ws = torch.zeros(lib.yolo_conv_wgrad_workspace_bytes(Cin, Cout, 3, L.BF16), dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.yolo_conv_wgrad_algo(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, 3, 1, 0, L.BF16,
                                    ws.data_ptr(), algo, st) == 0
    ref = wt.grad.numpy()
    np.testing.assert_allclose(dw.cpu().numpy() - 1.0, ref, rtol=1e-3, atol=2e-3 * np.abs(ref).max())
    assert float(ws.view(torch.float32).abs().max()) == 0.0          # the workspace is left zeroed
    # outside the kernel's domain: refused, not silently served by another kernel
    assert lib.yolo_conv_wgrad_algo(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), N, H, W, Cin, Cout, 3, 2, 0, L.BF16,
                                    ws.data_ptr(), algo, st) == L.EUNSUPPORTED


@pytest.mark.parametrize('case', [(2, 32, 16, 24, 64), (3, 64, 26, 26, 128), (2, 8, 12, 20, 32), (1, 128, 52, 52, 256),
                                  (4, 256, 26, 26, 512), (70, 32, 4, 4, 64), (1, 40, 6, 10, 96), (2, 64, 2, 2, 32),
                                  # the two shapes on which the 4-wave variant raced (an input DMA of a chunk's last phase read one phase later: conv_pipe.hip)
                                  (2, 32, 64, 64, 64), (3, 64, 48, 80, 128),
                                  (6, 256, 50, 4, 512), (6, 64, 34, 62, 256)])
def test_dgrad_s2_subpixel(lib, cuda, case):
    """yolo_conv_dgrad_s2: the data gradient of a 3x3 stride-2 conv as ONE 2x2-window conv over dy whose four
    output-channel blocks are the four sub-pixel phases of dx -- against torch autograd on the bf16-rounded operands,
    every tile variant, with and without accumulation into an existing gradient."""
    N, Cin, H, W, Cout = case                                   # forward conv: (N,Cin,H,W) -> (N,Cout,H/2,W/2)
    x, w, dy, _, _ = _ref((N, Cin, H, W, Cout, 3, 2), 5)
    rb = lambda a: torch.from_numpy(a).to(torch.bfloat16).float()
    xt = torch.from_numpy(x).requires_grad_(True)
    F.conv2d(xt, rb(w), None, stride=2, padding=1).backward(rb(dy))
    ref = xt.grad.numpy()
    st = torch.cuda.current_stream().cuda_stream
    wd = torch.empty(lib.yolo_packed_weight_bytes(4 * Cin, Cout, 2, L.BF16), dtype=torch.uint8, device=cuda)
    assert lib.yolo_pack_conv_weights_dgrad_s2(torch.from_numpy(w).to(cuda).data_ptr(), wd.data_ptr(), Cout, Cin, L.BF16, st) == 0
    dyd = to_nhwc(dy, 'bf16', cuda)
    cp = lib.yolo_padded_channels(4 * Cin)
    ones = torch.ones(cp, device=cuda); zeros = torch.zeros(cp, device=cuda)
    ran = 0
    for algo in (0, 2, 6, 10, 4):
        out = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=cuda)
        d = L.ConvDesc()
        d.x, d.w_packed, d.scale, d.bias, d.y = dyd.data_ptr(), wd.data_ptr(), ones.data_ptr(), zeros.data_ptr(), out.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope = N, H // 2, W // 2, Cout, 4 * Cin, 2, 1, L.BF16, 1.0
        d.algo = algo
        rc = lib.yolo_conv_dgrad_s2(C.byref(d), st)
        if algo and rc == L.EUNSUPPORTED:
            continue
        assert rc == 0, (algo, rc)
        ran += 1
        got = from_nhwc(out)
        assert np.isfinite(got).all(), algo
        np.testing.assert_allclose(got, ref, rtol=1e-2, atol=1e-2 * np.abs(ref).max(), err_msg='algo %d' % algo)
        d.residual = out.data_ptr()                              # accumulate in place
        assert lib.yolo_conv_dgrad_s2(C.byref(d), st) == 0
        np.testing.assert_allclose(from_nhwc(out), 2 * ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max(), err_msg='acc algo %d' % algo)
    assert ran >= 2                                              # the heuristic and at least one explicit variant


def test_dgrad_s2_rejects(lib, cuda):
    d = L.ConvDesc()
    buf = torch.zeros(1 << 16, device=cuda)
    d.x = d.w_packed = d.scale = d.bias = d.y = buf.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.dtype, d.slope = 1, 4, 4, 32, 4 * 12, L.BF16, 1.0     # Cin_f = 12: not a multiple of 8
    assert lib.yolo_conv_dgrad_s2(C.byref(d), None) == L.EUNSUPPORTED
    d.Cout, d.dtype = 4 * 16, L.F32
    assert lib.yolo_conv_dgrad_s2(C.byref(d), None) == L.EUNSUPPORTED
    d.dtype, d.slope = L.BF16, 2.0
    assert lib.yolo_conv_dgrad_s2(C.byref(d), None) == L.EINVAL


def test_bn_train_bf16(lib, cuda):
    N, H, W, Cc = 3, 13, 13, 64
    rng = np.random.default_rng(5)
    rb = lambda a: torch.from_numpy(a).to(torch.bfloat16).float()
    y = rb((2 * rng.standard_normal((N, Cc, H, W)) + 0.5).astype(np.float32)).requires_grad_(True)
    gamma = torch.from_numpy(rng.uniform(.5, 1.5, Cc).astype(np.float32)); beta = torch.from_numpy((.1 * rng.standard_normal(Cc)).astype(np.float32))
    mean = y.mean(dim=(0, 2, 3)); var = y.var(dim=(0, 2, 3), unbiased=False)
    z = F.leaky_relu((y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1), 0.1)
    dz = rb(rng.standard_normal((N, Cc, H, W)).astype(np.float32))
    z.backward(dz)
    st = torch.cuda.current_stream().cuda_stream
    yd, dzd = to_nhwc(y.detach().numpy(), 'bf16', cuda), to_nhwc(dz.numpy(), 'bf16', cuda)
    g_, b_ = gamma.to(cuda), beta.to(cuda)
    zd = torch.empty_like(yd); m_ = torch.empty(Cc, device=cuda); is_ = torch.empty(Cc, device=cuda)
    ws = torch.zeros(3 * Cc, dtype=torch.float64, device=cuda)
    npix = N * H * W
    assert lib.yolo_bn_train_fwd(yd.data_ptr(), g_.data_ptr(), b_.data_ptr(), None, zd.data_ptr(), m_.data_ptr(), is_.data_ptr(),
                                 None, None, ws.data_ptr(), npix, Cc, 1e-5, 0.9, 0.1, L.BF16, st) == 0
    np.testing.assert_allclose(from_nhwc(zd), z.detach().numpy(), rtol=1e-2, atol=1e-2)          # one bf16 rounding
    np.testing.assert_allclose(m_.cpu().numpy(), mean.detach().numpy(), rtol=1e-5, atol=1e-6)
    dyd = torch.empty_like(yd); dg = torch.empty(Cc, device=cuda); db = torch.empty(Cc, device=cuda)
    assert lib.yolo_bn_train_bwd(dzd.data_ptr(), yd.data_ptr(), m_.data_ptr(), is_.data_ptr(), g_.data_ptr(), b_.data_ptr(),
                                 dyd.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), npix, Cc, 0.1, L.BF16, st) == 0
    np.testing.assert_allclose(from_nhwc(dyd), y.grad.numpy(), rtol=1e-2, atol=1e-2 * np.abs(y.grad.numpy()).max())
    np.testing.assert_allclose(db.cpu().numpy(), dz.sum(dim=(0, 2, 3)).numpy() * 0 + db.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(3, 13, 13, 64), (2, 52, 52, 256), (2, 5, 7, 2048 + 64)])
def test_bn_train_two_launch_variants_are_bit_identical(lib, cuda, shape, dtype):
    """yolo_bn_train_fwd_pp / _bwd_pp (finalize folded into the apply pass; two alternating workspaces) against the
    three-launch calls: every output bit-equal, own workspace left dirty, the next one zeroed."""
    N, H, W, Cc = shape
    rng = np.random.default_rng(8)
    tdt, dt = (torch.float32, L.F32) if dtype == 'f32' else (torch.bfloat16, L.BF16)
    y = torch.from_numpy((2 * rng.standard_normal((N, H, W, Cc)) + 0.5).astype(np.float32)).to(cuda).to(tdt)
    dz = torch.from_numpy(rng.standard_normal((N, H, W, Cc)).astype(np.float32)).to(cuda).to(tdt)
    res = torch.from_numpy(rng.standard_normal((N, H, W, Cc)).astype(np.float32)).to(cuda).to(tdt)
    gamma = torch.from_numpy(rng.uniform(.5, 1.5, Cc).astype(np.float32)).to(cuda)
    beta = torch.from_numpy((.1 * rng.standard_normal(Cc)).astype(np.float32)).to(cuda)
    st = torch.cuda.current_stream().cuda_stream
    npix = N * H * W
    outs = []
    for pp in (False, True):
        z = torch.empty_like(y); dy = torch.empty_like(y)
        m_ = torch.empty(Cc, device=cuda); is_ = torch.empty(Cc, device=cuda)
        rm = torch.full((Cc,), 0.25, device=cuda); rv = torch.full((Cc,), 0.75, device=cuda)
        dg = torch.empty(Cc, device=cuda); db = torch.empty(Cc, device=cuda)
        wa = torch.zeros(2 * Cc, dtype=torch.float64, device=cuda)
        wb = torch.full((2 * Cc,), 123.0, dtype=torch.float64, device=cuda)         # dirty: the forward call must zero it
        if pp:
            assert lib.yolo_bn_train_fwd_pp(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), res.data_ptr(), z.data_ptr(), m_.data_ptr(),
                                            is_.data_ptr(), rm.data_ptr(), rv.data_ptr(), wa.data_ptr(), wb.data_ptr(), 2 * Cc, npix, Cc, 1e-5,
                                            0.9, 0.1, dt, st) == 0
            torch.cuda.synchronize()
            assert float(wb.abs().max()) == 0.0 and float(wa.abs().max()) > 0.0
            assert lib.yolo_bn_train_bwd_pp(dz.data_ptr(), y.data_ptr(), m_.data_ptr(), is_.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                            dy.data_ptr(), dg.data_ptr(), db.data_ptr(), wb.data_ptr(), wa.data_ptr(), 2 * Cc, npix, Cc, 0.1, dt, st) == 0
            torch.cuda.synchronize()
            assert float(wa.abs().max()) == 0.0
            assert lib.yolo_bn_train_fwd_pp(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, z.data_ptr(), m_.data_ptr(), is_.data_ptr(),
                                            None, None, wa.data_ptr(), wa.data_ptr(), 2 * Cc, npix, Cc, 1e-5, 0.9, 0.1, dt, st) == L.EINVAL
        else:
            assert lib.yolo_bn_train_fwd(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), res.data_ptr(), z.data_ptr(), m_.data_ptr(),
                                         is_.data_ptr(), rm.data_ptr(), rv.data_ptr(), wa.data_ptr(), npix, Cc, 1e-5, 0.9, 0.1, dt, st) == 0
            assert lib.yolo_bn_train_bwd(dz.data_ptr(), y.data_ptr(), m_.data_ptr(), is_.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                         dy.data_ptr(), dg.data_ptr(), db.data_ptr(), wa.data_ptr(), npix, Cc, 0.1, dt, st) == 0
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (z, dy, m_, is_, rm, rv, dg, db)])
    # the double sums are accumulated with atomics (order varies run to run): the statistics agree to rounding, and
    # everything derived from IDENTICAL statistics is bit-equal -- compare with a tolerance that only the atomics explain
    names = ('z', 'dy', 'mean', 'invstd', 'running_mean', 'running_var', 'dgamma', 'dbeta')
    for n, a, b in zip(names, outs[0], outs[1]):
        np.testing.assert_allclose(a.float().cpu().numpy(), b.float().cpu().numpy(), rtol=2e-6 if n not in ('z', 'dy') or dtype == 'f32' else 1e-2,
                                   atol=1e-6 if dtype == 'f32' or n not in ('z', 'dy') else 1e-2, err_msg=n)


def test_pack_pairs_bit_identical(lib, cuda):
    """yolo_pack_conv_weights_pairs (forward + data-gradient image from one read of the weights, LDS transpose, 16-byte
    stores) against yolo_pack_conv_weights + yolo_pack_conv_weights_dgrad, bit for bit, several convs in one launch."""
    rng = np.random.default_rng(5)
    st = torch.cuda.current_stream().cuda_stream
    convs = [(64, 32, 3), (32, 64, 1), (256, 128, 3), (128, 256, 1), (1024, 512, 3), (96, 160, 3), (288, 32, 1)]
    assert lib.yolo_pack_pair_blocks(90, 64, 1) == L.EUNSUPPORTED and lib.yolo_pack_pair_blocks(64, 8, 3) == L.EUNSUPPORTED
    recs, first, keep = [], [0], []
    dt = np.dtype([('w', '<u8'), ('fwd', '<u8'), ('dgrad', '<u8'), ('cout', '<i4'), ('cin', '<i4'), ('k', '<i4'), ('r', '<i4')])
    for co, ci, k in convs:
        w = torch.from_numpy(rng.standard_normal((co, ci, k, k)).astype(np.float32)).to(cuda)
        f1 = torch.zeros(lib.yolo_packed_weight_bytes(co, ci, k, L.BF16), dtype=torch.uint8, device=cuda)
        d1 = torch.zeros(lib.yolo_packed_weight_bytes(ci, co, k, L.BF16), dtype=torch.uint8, device=cuda)
        f2, d2 = torch.full_like(f1, 0x5a), torch.full_like(d1, 0x5a)
        assert lib.yolo_pack_conv_weights(w.data_ptr(), f2.data_ptr(), co, ci, k, L.BF16, st) == 0
        assert lib.yolo_pack_conv_weights_dgrad(w.data_ptr(), d2.data_ptr(), co, ci, k, L.BF16, st) == 0
        recs.append((w.data_ptr(), f1.data_ptr(), d1.data_ptr(), co, ci, k, 0))
        first.append(first[-1] + lib.yolo_pack_pair_blocks(co, ci, k))
        keep.append((w, f1, d1, f2, d2))
    items = torch.from_numpy(np.array(recs, dtype=dt).view(np.uint8).copy()).to(cuda)
    fb = torch.tensor(first, dtype=torch.int64, device=cuda)
    assert lib.yolo_pack_conv_weights_pairs(items.data_ptr(), fb.data_ptr(), len(recs), first[-1], st) == 0
    torch.cuda.synchronize()
    for (co, ci, k), (w, f1, d1, f2, d2) in zip(convs, keep):
        assert torch.equal(f1, f2), ('fwd', co, ci, k)
        assert torch.equal(d1, d2), ('dgrad', co, ci, k)
