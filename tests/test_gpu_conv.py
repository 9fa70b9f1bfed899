"""GPU parity of the implicit-GEMM conv kernel (through the C ABI) against a torch-CPU fp32
restatement of Convolution + folded BatchNorm + LeakyReLU (+ residual)."""
import numpy as np
import pytest

import torch
from util import run_conv, ref_conv, to_nhwc, from_nhwc, eligible_pairs, conv_variant
from yolo_amd import lib as L

pytestmark = pytest.mark.gpu

# (N, Cin, H, W, Cout, k, stride, residual)
CASES = [
    (2, 8, 16, 24, 16, 3, 1, False),     # stem-like: Cin=8 (one partial K-chunk)
    (2, 16, 16, 24, 32, 3, 2, False),    # stride-2 down-sample
    (2, 32, 13, 13, 64, 3, 1, True),     # odd map, residual, batch-crossing tiles
    (3, 64, 13, 13, 32, 1, 1, False),    # 1x1, TS=2 (bf16) / TS=4 (f32)
    (2, 128, 26, 26, 256, 3, 1, True),   # 128x128 tile config, 4 K-chunks
    (1, 256, 13, 13, 128, 1, 1, False),  # 1x1 with 8 chunks (bf16) -> TS=4
    (2, 64, 26, 26, 128, 3, 2, False),   # stride 2 -> 13x13, 128-wide tile
    (2, 96, 8, 8, 88, 1, 1, False),      # ragged Cout (not a tile multiple), Cin=96
    (5, 16, 3, 5, 24, 3, 1, False),      # tiny map: many images per tile
    (1, 32, 52, 52, 64, 3, 1, True),     # multi-strip? (Wo=52)
    (1, 16, 40, 104, 32, 3, 1, False),   # wide map: several strips per row
]


def _mk(case, seed):
    N, Cin, H, W, Cout, k, stride, res = case
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    r = rng.standard_normal((N, Cout, Ho, Wo)).astype(np.float32) if res else None
    return x, w, scale, bias, r


@pytest.mark.parametrize('case', CASES)
def test_conv_f32(lib, cuda, case):
    x, w, scale, bias, r = _mk(case, 1)
    y = run_conv(lib, cuda, x, w, scale, bias, case[6], 0.1, 'f32', residual=r)
    ref = ref_conv(x, w, scale, bias, case[6], 0.1, residual=r)
    assert not np.isnan(y).any()
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)      # fp32 path: accumulation-order noise only


@pytest.mark.parametrize('case', CASES)
def test_conv_bf16(lib, cuda, case):
    x, w, scale, bias, r = _mk(case, 2)
    y = run_conv(lib, cuda, x, w, scale, bias, case[6], 0.1, 'bf16', residual=r)
    ref = ref_conv(x, w, scale, bias, case[6], 0.1, residual=r, bf16=True)
    assert not np.isnan(y).any()
    # same bf16-rounded operands, fp32 accumulate: differences are one bf16 ulp of the output at most
    np.testing.assert_allclose(y, ref, rtol=1.6e-2, atol=2e-2)
    assert np.mean(np.abs(y - ref) > 1e-3 * (1 + np.abs(ref))) < 0.02


@pytest.mark.parametrize('case', CASES)
def test_conv_f16(lib, cuda, case):
    """YOLO_F16 (the reference's use_fp16, car/YOLO.py:98-100): the same kernels on v_mfma_f32_32x32x16_f16, operands and the
    stored activation rounded to IEEE half (v_cvt_pk_f16_f32, round-to-nearest-even) -- one half ulp of the output at most."""
    x, w, scale, bias, r = _mk(case, 2)
    y = run_conv(lib, cuda, x, w, scale, bias, case[6], 0.1, 'f16', residual=r)
    ref = ref_conv(x, w, scale, bias, case[6], 0.1, residual=r, bf16='f16')
    assert not np.isnan(y).any()
    np.testing.assert_allclose(y, ref, rtol=2e-3, atol=2e-3)
    assert np.mean(np.abs(y - ref) > 2e-4 * (1 + np.abs(ref))) < 0.02


# shapes no width-dividing strip fits (prime widths on tall batches, maps one or two pixels wide): the generic kernel's ragged
# strips / row-limited tiles (launch_cfg's fallback; these were refused with -2 until round 3 -- found by fuzzing the C ABI)
RAGGED = [(5, 16, 25, 61, 48, 3, 1, False), (5, 512, 38, 61, 384, 3, 1, False), (5, 320, 33, 67, 256, 3, 1, True),
          (5, 16, 39, 1, 64, 3, 1, False), (5, 256, 34, 61, 8, 3, 1, True), (4, 32, 31, 122, 64, 3, 2, False),
          (6, 24, 29, 2, 32, 3, 1, True), (5, 96, 31, 61, 256, 3, 1, False)]


@pytest.mark.parametrize('algo', [0, 1])
@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('case', RAGGED)
def test_conv_widths_no_strip_divides(lib, cuda, case, dtype, algo):
    x, w, scale, bias, r = _mk(case, 5)
    y = run_conv(lib, cuda, x, w, scale, bias, case[6], 0.1, dtype, residual=r, algo=algo)
    ref = ref_conv(x, w, scale, bias, case[6], 0.1, residual=r, bf16=(dtype == 'bf16'))
    assert not np.isnan(y).any()
    if dtype == 'f32':
        np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)
    else:
        np.testing.assert_allclose(y, ref, rtol=1.6e-2, atol=2e-2)


def test_conv_out_f32_linear(lib, cuda):
    """Head-logit mode: bias only, linear, float32 output from bf16 activations."""
    case = (2, 64, 13, 13, 90, 1, 1, False)
    x, w, scale, bias, _ = _mk(case, 3)
    scale[:] = 1.0
    y = run_conv(lib, cuda, x, w, scale, bias, 1, 1.0, 'bf16', out_f32=True)
    rb = lambda a: np.asarray(__import__('torch').from_numpy(a).to(__import__('torch').bfloat16).float())
    ref = ref_conv(rb(x), rb(w), scale, bias, 1, 1.0)
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)


# ---- pipelined variants (csrc/conv_pipe.hip): every algo id on shapes it is eligible for ----------
PIPE_CASES = [
    (2, 128, 26, 26, 256, 3, 1, True),
    (4, 64, 52, 52, 256, 3, 1, True),      # several strips per row
    (8, 256, 13, 13, 512, 3, 1, True),     # tiles crossing image boundaries
    (3, 128, 19, 19, 256, 3, 1, False),    # 608-family odd map
    (1, 64, 13, 13, 96, 3, 1, False),      # fewer pixels than one tile, ragged Cout
    (1, 256, 13, 13, 128, 1, 1, False),
    (4, 256, 26, 26, 512, 1, 1, True),
    (2, 512, 13, 13, 88, 1, 1, False),
    (2, 64, 26, 26, 128, 3, 1, False),     # Cout smaller than the widest cout tile
    (2, 32, 40, 40, 64, 3, 1, True),       # a single K-chunk in bf16 (9 phases)
]


PIPE_ALGOS = [2, 3, 4, 5, 6, 7, 8, 11, 12, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 30, 31, 32, 33, 35, 36, 37, 38, 39]


# only the (shape, dtype, variant) pairs the library accepts (tests/util.py:eligible_pairs; the refusal rules are asserted
# on the CPU by tests/test_host.py::test_conv_variant_eligibility_rules)
@pytest.mark.parametrize('case,dtype,algo', eligible_pairs(PIPE_CASES, ['f32', 'bf16', 'f16'], PIPE_ALGOS))
def test_conv_pipe(lib, cuda, case, dtype, algo):
    x, w, scale, bias, r = _mk(case, 4)
    y = run_conv(lib, cuda, x, w, scale, bias, case[6], 0.1, dtype, residual=r, algo=algo, expect_rc=0)
    ref = ref_conv(x, w, scale, bias, case[6], 0.1, residual=r, bf16={'f32': False, 'bf16': True, 'f16': 'f16'}[dtype])
    assert not np.isnan(y).any()
    if dtype == 'f32':
        np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)
    elif dtype == 'f16':
        np.testing.assert_allclose(y, ref, rtol=2e-3, atol=2e-3)
        assert np.mean(np.abs(y - ref) > 2e-4 * (1 + np.abs(ref))) < 0.02
    else:
        np.testing.assert_allclose(y, ref, rtol=1.6e-2, atol=2e-2)
        assert np.mean(np.abs(y - ref) > 1e-3 * (1 + np.abs(ref))) < 0.02


# prime widths (one strip as wide as the map, or refusal): the tiles whose conflict-free halo (pitch TWt + 4, conv_pipe.hip
# launch_pipe) does not fit their buffer fall back to the plain pitch TWt + 2, others take the padded one; images of 9 / 21 rows put
# fragments across image boundaries in both layouts
LAYOUT_CASES = [(2, 64, 9, 31, 128, 3, 1, True), (3, 32, 21, 37, 64, 3, 1, False), (2, 64, 5, 43, 128, 3, 1, True), (1, 128, 30, 47, 64, 3, 1, False)]


@pytest.mark.parametrize('case,dtype,algo', eligible_pairs(LAYOUT_CASES, ['f32', 'bf16'], [2, 3, 4, 5, 6, 7, 8, 11, 26, 27, 28]))
def test_conv_pipe_halo_layouts(lib, cuda, case, dtype, algo):
    x, w, scale, bias, r = _mk(case, 14)
    y = run_conv(lib, cuda, x, w, scale, bias, 1, 0.1, dtype, residual=r, algo=algo, expect_rc=0)
    ref = ref_conv(x, w, scale, bias, 1, 0.1, residual=r, bf16=dtype == 'bf16')
    assert not np.isnan(y).any()
    if dtype == 'f32':
        np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)
    else:
        np.testing.assert_allclose(y, ref, rtol=1.6e-2, atol=2e-2)
        assert np.mean(np.abs(y - ref) > 1e-3 * (1 + np.abs(ref))) < 0.02


def test_conv_pipe_rejects_ineligible(lib, cuda):
    """A pinned pipelined algo on a shape it cannot run must fail loudly, not fall back."""
    case = (2, 16, 16, 24, 32, 3, 2, False)          # stride 2
    x, w, scale, bias, r = _mk(case, 5)
    run_conv(lib, cuda, x, w, scale, bias, 2, 0.1, 'bf16', algo=2, expect_rc=-2)


S2_CASES = [
    (2, 64, 26, 26, 128, 3, 2, False),      # -> 13x13
    (4, 128, 52, 52, 256, 3, 2, False),     # -> 26x26, tiles crossing images
    (3, 64, 38, 38, 96, 3, 2, False),       # 608 family -> 19x19, ragged Cout
    (2, 128, 16, 40, 64, 3, 2, False),
    (2, 64, 22, 58, 128, 3, 2, False),      # -> 11x29 (a prime width: the padded pitch of the conflict-free layout or the plain one)
]


S2_ALGOS = [9, 10, 16, 17, 18]


@pytest.mark.parametrize('case,dtype,algo', eligible_pairs(S2_CASES, ['f32', 'bf16', 'f16'], S2_ALGOS))
def test_conv_pipe_stride2(lib, cuda, case, dtype, algo):
    x, w, scale, bias, r = _mk(case, 6)
    y = run_conv(lib, cuda, x, w, scale, bias, 2, 0.1, dtype, algo=algo, expect_rc=0)
    ref = ref_conv(x, w, scale, bias, 2, 0.1, bf16={'f32': False, 'bf16': True, 'f16': 'f16'}[dtype])
    assert not np.isnan(y).any()
    if dtype == 'f32':
        np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)
    elif dtype == 'f16':
        np.testing.assert_allclose(y, ref, rtol=2e-3, atol=2e-3)
    else:
        np.testing.assert_allclose(y, ref, rtol=1.6e-2, atol=2e-2)


# ---- streaming kernel for the small-channel layers (csrc/conv_stream.hip, algo 13 / 14) -----------------------
STREAM_CASES = [
    # (N, Cin, H, W, Cout, k, stride, residual)
    (2, 32, 40, 75, 64, 3, 1, True),        # several balanced strips, ragged last strip, row slices
    (1, 32, 9, 200, 64, 3, 1, False),
    (3, 64, 33, 41, 128, 3, 1, True),
    (2, 64, 20, 70, 32, 3, 1, True),        # data-gradient shape of a residual block's 3x3
    (2, 64, 24, 24, 64, 3, 1, False),
    (2, 32, 41, 77, 64, 3, 2, False),       # stride 2, odd input size
    (2, 64, 38, 38, 128, 3, 2, False),
    (2, 64, 30, 131, 32, 1, 1, False),
    (3, 128, 19, 19, 64, 1, 1, False),
    (2, 32, 17, 140, 64, 1, 1, True),       # data-gradient shape of a residual block's 1x1
    (2, 64, 26, 26, 128, 1, 1, True),
    (1, 128, 13, 70, 128, 1, 1, False),
]


@pytest.mark.parametrize('case,dtype,algo', eligible_pairs(STREAM_CASES, ['bf16', 'f16'], [13, 14]))
def test_conv_stream(lib, cuda, case, dtype, algo):
    x, w, scale, bias, r = _mk(case, 7)
    k16 = 1.0 if dtype == 'bf16' else 0.125
    y = run_conv(lib, cuda, x, w, scale, bias, case[6], 0.1, dtype, residual=r, algo=algo, expect_rc=0)
    ref = ref_conv(x, w, scale, bias, case[6], 0.1, residual=r, bf16=(True if dtype == 'bf16' else 'f16'))
    assert not np.isnan(y).any()
    np.testing.assert_allclose(y, ref, rtol=1.6e-2 * k16, atol=2e-2 * k16)
    assert np.mean(np.abs(y - ref) > 1e-3 * k16 * (1 + np.abs(ref))) < 0.02


def test_conv_stream_rejects_ineligible(lib, cuda):
    case = (2, 256, 13, 13, 128, 1, 1, False)
    x, w, scale, bias, r = _mk(case, 8)
    run_conv(lib, cuda, x, w, scale, bias, 1, 0.1, 'bf16', algo=13, expect_rc=-2)
    run_conv(lib, cuda, x[:, :64], w[:, :64], scale, bias, 1, 0.1, 'f32', algo=13, expect_rc=-2)


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 64, 96), (1, 37, 61), (3, 8, 8), (1, 200, 250), (2, 24, 248), (1, 2, 2), (5, 33, 130)])
def test_stem_down_fused_is_bit_identical(lib, cuda, shape, dtype):
    """yolo_stem_down_fwd (stem 3->32 + first down-sampling conv 32->64 in one kernel, the 32-channel map kept in LDS)
    against the two separate kernels it replaces: same operand / accumulation order and rounding points, so the outputs
    must be bit-identical -- odd sizes, strips at the 62-pixel limit, several row slices, and against the oracle."""
    import ctypes as C
    import torch
    from yolo_amd import lib as L
    from util import TDT, LDT
    ldt, tdt = LDT[dtype], TDT[dtype]
    N, H, W = shape
    rng = np.random.default_rng(11)
    x = rng.random((N, 3, H, W)).astype(np.float32)
    w1 = (rng.standard_normal((32, 3, 3, 3)) / 5).astype(np.float32)
    w2 = (rng.standard_normal((64, 32, 3, 3)) / 17).astype(np.float32)
    s1, b1 = rng.uniform(0.5, 1.5, 32).astype(np.float32), rng.uniform(-0.5, 0.5, 32).astype(np.float32)
    s2, b2 = rng.uniform(0.5, 1.5, 64).astype(np.float32), rng.uniform(-0.5, 0.5, 64).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    dev = cuda
    xd = torch.from_numpy(x).to(dev)
    w1d, w2d = torch.from_numpy(w1).to(dev), torch.from_numpy(w2).to(dev)
    wp2 = torch.empty(lib.yolo_packed_weight_bytes(64, 32, 3, ldt), dtype=torch.uint8, device=dev)
    assert lib.yolo_pack_conv_weights(w2d.data_ptr(), wp2.data_ptr(), 64, 32, 3, ldt, st) == 0
    pad = lambda v: torch.cat([torch.from_numpy(v), torch.zeros(lib.yolo_padded_channels(len(v)) - len(v))]).to(dev)
    s1d, b1d, s2d, b2d = pad(s1), pad(b1), pad(s2), pad(b2)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    # the two-kernel path
    mid = torch.empty((N, H, W, 32), dtype=tdt, device=dev)
    assert lib.yolo_stem_conv_fwd(xd.data_ptr(), w1d.data_ptr(), s1d.data_ptr(), b1d.data_ptr(), mid.data_ptr(), N, H, W, 3, 32,
                                  ldt, 0.1, st) == 0
    two = torch.full((N, Ho, Wo, 64), float('nan'), dtype=tdt, device=dev)
    d = L.ConvDesc()
    d.x, d.w_packed, d.scale, d.bias, d.y = mid.data_ptr(), wp2.data_ptr(), s2d.data_ptr(), b2d.data_ptr(), two.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope = N, H, W, 32, 64, 3, 2, ldt, 0.1
    assert lib.yolo_conv_fwd(C.byref(d), st) == 0
    one = torch.full((N, Ho, Wo, 64), float('nan'), dtype=tdt, device=dev)
    assert lib.yolo_stem_down_fwd(xd.data_ptr(), w1d.data_ptr(), s1d.data_ptr(), b1d.data_ptr(), wp2.data_ptr(), s2d.data_ptr(),
                                  b2d.data_ptr(), one.data_ptr(), N, H, W, 32, 64, ldt, 0.1, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(one.view(torch.int16), two.view(torch.int16))
    # and the oracle of the pair (rounding-aware)
    r1 = ref_conv(x, w1, s1, b1, 1, 0.1, bf16=(True if dtype == 'bf16' else 'f16'))
    r2 = ref_conv(r1, w2, s2, b2, 2, 0.1, bf16=(True if dtype == 'bf16' else 'f16'))
    got = one.float().permute(0, 3, 1, 2).cpu().numpy()
    tol = 2e-2 if dtype == 'bf16' else 3e-3
    np.testing.assert_allclose(got, r2, rtol=tol, atol=tol * np.abs(r2).max())


def test_stem_down_rejects(lib, cuda):
    import torch
    from yolo_amd import lib as L
    b = torch.zeros(4096, device=cuda)
    p = b.data_ptr()
    args = lambda c1, c2, dt, slope: (p, p, p, p, p, p, p, p, 1, 8, 8, c1, c2, dt, slope, None)
    assert lib.yolo_stem_down_fwd(*args(16, 64, L.BF16, 0.1)) == L.EUNSUPPORTED
    assert lib.yolo_stem_down_fwd(*args(32, 128, L.BF16, 0.1)) == L.EUNSUPPORTED
    assert lib.yolo_stem_down_fwd(*args(32, 64, L.F32, 0.1)) == L.EUNSUPPORTED
    assert lib.yolo_stem_down_fwd(*args(32, 64, L.BF16, 1.5)) == L.EINVAL


IDENT_CASES = [((2, 128, 26, 26, 256, 3, 1, True), 0), ((2, 128, 26, 26, 256, 3, 1, True), 4),
               ((3, 256, 13, 13, 88, 1, 1, False), 1), ((2, 64, 40, 40, 128, 3, 1, True), 13),
               ((2, 64, 26, 26, 128, 3, 2, False), 0), ((1, 256, 13, 13, 512, 1, 1, True), 11)]


@pytest.mark.parametrize('case,algo,dtype', [(c, a, dt) for c, a in IDENT_CASES for dt in ('bf16', 'f32') if conv_variant(c, dt, a) is not None])
def test_identity_epilogue(lib, cuda, case, algo, dtype):
    """scale == bias == NULL (the training step's raw convolutions / data gradients): bit-identical to scale 1, bias 0,
    slope 1 through every epilogue path; exactly one of the two NULL is an argument error."""
    import ctypes as C
    import torch
    from yolo_amd import lib as L
    from util import to_nhwc, LDT, TDT
    N, Cin, H, W, Cout, k, stride, with_res = case
    x, w, _, _, r = _mk(case, 21)
    st = torch.cuda.current_stream().cuda_stream
    dt = LDT[dtype]
    xd = to_nhwc(x, dtype, cuda)
    wp = torch.empty(lib.yolo_packed_weight_bytes(Cout, Cin, k, dt), dtype=torch.uint8, device=cuda)
    L.check(lib.yolo_pack_conv_weights(torch.from_numpy(w).to(cuda).data_ptr(), wp.data_ptr(), Cout, Cin, k, dt, st), 'pack')
    cp = lib.yolo_padded_channels(Cout)
    ones, zeros = torch.ones(cp, device=cuda), torch.zeros(cp, device=cuda)
    rd = to_nhwc(r, dtype, cuda) if r is not None else None
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    outs = []
    for sc, bi in ((ones.data_ptr(), zeros.data_ptr()), (None, None)):
        y = torch.full((N, Ho, Wo, Cout), float('nan'), dtype=TDT[dtype], device=cuda)
        d = L.ConvDesc()
        d.x, d.w_packed, d.scale, d.bias, d.y = xd.data_ptr(), wp.data_ptr(), sc, bi, y.data_ptr()
        d.residual = rd.data_ptr() if rd is not None else None
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = N, H, W, Cin, Cout, k, stride, dt, 1.0, algo
        assert lib.yolo_conv_fwd(C.byref(d), st) == 0
        outs.append(y)
    torch.cuda.synchronize()
    it = torch.int16 if dtype == 'bf16' else torch.int32
    assert not torch.isnan(outs[1].float()).any()
    assert torch.equal(outs[0].view(it), outs[1].view(it))
    d.scale = ones.data_ptr()
    assert lib.yolo_conv_fwd(C.byref(d), st) == L.EINVAL


@pytest.mark.parametrize('algo', [2, 3, 4, 5, 8, 11, 12])
@pytest.mark.parametrize('cin', [64, 96, 128])
def test_conv_pipe_1x1_short_k(lib, cuda, algo, cin):
    """1x1 variants on K extents below four phases (bf16: 2-3 chunks of 32 channels) take the generic K loop instead of
    the lean one, and exactly four phases is the lean loop's shortest case (its whole body is the tail)."""
    case = (2, cin, 13, 13, 128, 1, 1, True)
    x, w, scale, bias, r = _mk(case, 5)
    y = run_conv(lib, cuda, x, w, scale, bias, 1, 0.1, 'bf16', residual=r, algo=algo)
    ref = ref_conv(x, w, scale, bias, 1, 0.1, residual=r, bf16=True)
    assert not np.isnan(y).any()
    np.testing.assert_allclose(y, ref, rtol=1.6e-2, atol=2e-2)


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('case', [(2, 37, 70, 64), (1, 20, 130, 128), (3, 5, 62, 64), (1, 40, 8, 128), (2, 52, 104, 128),
                                  (1, 33, 63, 64), (1, 64, 208, 64)])
def test_res_block_fused(lib, cuda, case, dtype):
    """yolo_res_block_fwd (one DarknetBasicBlockV3 of the first stages as one kernel) against the torch reference on the
    bf16-rounded operands with the HIP path's rounding points, and against the two separate HIP layers: several strips
    (balanced, ragged), several row slices, image edges on every side."""
    import ctypes as C
    from util import TDT, LDT
    ldt, tdt, sim = LDT[dtype], TDT[dtype], (True if dtype == 'bf16' else 'f16')
    k16 = 1.0 if dtype == 'bf16' else 0.125           # (half: 3 more mantissa bits)
    N, H, W, Cc = case
    rng = np.random.default_rng(21)
    x = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    w1 = (rng.standard_normal((Cc // 2, Cc, 1, 1)) / np.sqrt(Cc)).astype(np.float32)
    w2 = (rng.standard_normal((Cc, Cc // 2, 3, 3)) / np.sqrt(Cc // 2 * 9)).astype(np.float32)
    s1, b1 = rng.uniform(.5, 1.5, Cc // 2).astype(np.float32), (.3 * rng.standard_normal(Cc // 2)).astype(np.float32)
    s2, b2 = rng.uniform(.5, 1.5, Cc).astype(np.float32), (.3 * rng.standard_normal(Cc)).astype(np.float32)
    mid = ref_conv(x, w1, s1, b1, 1, 0.1, bf16=sim)
    ref = ref_conv(mid, w2, s2, b2, 1, 0.1, residual=x, bf16=sim)
    st = torch.cuda.current_stream().cuda_stream

    def pack(w):
        co, ci, k, _ = w.shape
        wp = torch.empty(lib.yolo_packed_weight_bytes(co, ci, k, ldt), dtype=torch.uint8, device=cuda)
        L.check(lib.yolo_pack_conv_weights(torch.from_numpy(w).to(cuda).data_ptr(), wp.data_ptr(), co, ci, k, ldt, st), 'pack')
        return wp

    def padded(v):
        t = torch.zeros(lib.yolo_padded_channels(len(v)), device=cuda)
        t[:len(v)] = torch.from_numpy(v).to(cuda)
        return t

    wp1, wp2 = pack(w1), pack(w2)
    ts1, tb1, ts2, tb2 = padded(s1), padded(b1), padded(s2), padded(b2)
    xd = to_nhwc(x, dtype, cuda)
    y = torch.full((N, H, W, Cc), float('nan'), dtype=tdt, device=cuda)
    rc = lib.yolo_res_block_fwd(xd.data_ptr(), wp1.data_ptr(), ts1.data_ptr(), tb1.data_ptr(), wp2.data_ptr(), ts2.data_ptr(),
                                tb2.data_ptr(), y.data_ptr(), N, H, W, Cc, ldt, 0.1, st)
    assert rc == 0
    torch.cuda.synchronize()
    got = from_nhwc(y)
    assert np.isfinite(got).all()
    # one bf16 ulp of the output (the mid map may differ from the reference's by one rounding, too)
    np.testing.assert_allclose(got, ref, rtol=2e-2 * k16, atol=3e-2 * k16)
    # (half: a finer mid map flips 8x more often against the reference's, each flip 8x smaller -- the MEAN error does not scale with
    #  the ulp the way the maximum does)
    assert np.abs(got - ref).mean() < (2e-3 if dtype == 'bf16' else 5e-4)
    # the two separate layers through yolo_conv_fwd: same operands and rounding points -> equal up to fp32 summation order
    mid_h = run_conv(lib, cuda, x, w1, s1, b1, 1, 0.1, dtype)
    sep = run_conv(lib, cuda, mid_h, w2, s2, b2, 1, 0.1, dtype, residual=x)
    assert np.mean(got != sep) < 2e-3 and np.abs(got - sep).max() < 0.07 * k16


def test_res_block_rejects(lib, cuda):
    buf = torch.zeros(1 << 16, device=cuda)
    p = buf.data_ptr()
    assert lib.yolo_res_block_fwd(p, p, p, p, p, p, p, p, 1, 8, 8, 256, L.BF16, 0.1, None) == L.EUNSUPPORTED
    assert lib.yolo_res_block_fwd(p, p, p, p, p, p, p, p, 1, 8, 8, 64, L.F32, 0.1, None) == L.EUNSUPPORTED
    assert lib.yolo_res_block_fwd(p, None, p, p, p, p, p, p, 1, 8, 8, 64, L.BF16, 0.1, None) == L.EINVAL
    assert lib.yolo_res_block_fwd(p, p, p, p, p, p, p, p, 1, 8, 8, 64, L.BF16, 1.5, None) == L.EINVAL


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 64, 13, 13, 32, 1), (1, 128, 10, 14, 64, 1), (3, 32, 6, 5, 64, 3), (2, 512, 13, 13, 256, 1)])
def test_conv_strided_input_and_upsampled_output(lib, cuda, dtype, case):
    """yolo_conv_desc.x_pixel_stride (x = channel slice of a wider NHWC buffer) and .upsample2x (every output pixel stored
    to its 2x2 patch of a (N,2Ho,2Wo,*) buffer, into a channel slice): concat(upsample(conv(x)), route) of
    car/utils.py:91-93 without a copy.  Against the dense path on the same operands, every tile variant that takes the shape."""
    import ctypes as C
    N, Cin, H, W, Cout, k = case
    rng = np.random.default_rng(31)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    sc, bi = rng.uniform(.5, 1.5, Cout).astype(np.float32), (.2 * rng.standard_normal(Cout)).astype(np.float32)
    dense = run_conv(lib, cuda, x, w, sc, bi, 1, 0.1, dtype)                     # (N,Cout,H,W) through the plain path
    st = torch.cuda.current_stream().cuda_stream
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    dt = L.F32 if dtype == 'f32' else L.BF16
    # x lives in channels [40, 40+Cin) of a wider buffer; y goes to channels [8, 8+Cout) of a (N,2H,2W,Cout+24) buffer
    XC, x0, YC, y0 = Cin + 56, 40, Cout + 24, 8
    xbuf = torch.full((N, H, W, XC), float('nan'), dtype=tdt, device=cuda)
    xbuf[..., x0:x0 + Cin] = to_nhwc(x, dtype, cuda)
    wp = torch.empty(lib.yolo_packed_weight_bytes(Cout, Cin, k, dt), dtype=torch.uint8, device=cuda)
    L.check(lib.yolo_pack_conv_weights(torch.from_numpy(w).to(cuda).data_ptr(), wp.data_ptr(), Cout, Cin, k, dt, st), 'pack')
    cp = lib.yolo_padded_channels(Cout)
    scd = torch.zeros(cp, device=cuda); scd[:Cout] = torch.from_numpy(sc).to(cuda)
    bid = torch.zeros(cp, device=cuda); bid[:Cout] = torch.from_numpy(bi).to(cuda)
    ran = 0
    for algo in (0, 1, 2, 4, 8, 11, 22):
        ybuf = torch.full((N, 2 * H, 2 * W, YC), -7.0, dtype=tdt, device=cuda)
        d = L.ConvDesc()
        d.x, d.w_packed, d.scale, d.bias = xbuf[..., x0:].data_ptr(), wp.data_ptr(), scd.data_ptr(), bid.data_ptr()
        d.y = ybuf[..., y0:].data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = N, H, W, Cin, Cout, k, 1, dt, 0.1, algo
        d.x_pixel_stride, d.upsample2x, d.y_pixel_stride, d.y_batch_stride = XC, 1, YC, 4 * H * W * YC
        rc = lib.yolo_conv_fwd(C.byref(d), st)
        if algo and rc == L.EUNSUPPORTED:
            continue
        assert rc == 0, (algo, rc)
        ran += 1
        torch.cuda.synchronize()
        got = ybuf.float().cpu().numpy()
        assert (got[..., :y0] == -7.0).all() and (got[..., y0 + Cout:] == -7.0).all(), algo     # neighbours untouched
        up = np.repeat(np.repeat(dense, 2, axis=2), 2, axis=3)                                  # nearest 2x, NCHW
        np.testing.assert_array_equal(got[..., y0:y0 + Cout].transpose(0, 3, 1, 2), up, err_msg='algo %d' % algo)
    assert ran >= 2
    # strided output WITH a dense residual (a stage's last block writing its route half of the concat buffer)
    if Cout % 8 == 0:
        res = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
        ref = run_conv(lib, cuda, x, w, sc, bi, 1, 0.1, dtype, residual=res)
        rd = to_nhwc(res, dtype, cuda)
        ybuf = torch.full((N, H, W, YC), -7.0, dtype=tdt, device=cuda)
        d = L.ConvDesc()
        d.x, d.w_packed, d.scale, d.bias = xbuf[..., x0:].data_ptr(), wp.data_ptr(), scd.data_ptr(), bid.data_ptr()
        d.residual, d.y = rd.data_ptr(), ybuf[..., y0:].data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = N, H, W, Cin, Cout, k, 1, dt, 0.1, 0
        d.x_pixel_stride, d.y_pixel_stride, d.y_batch_stride = XC, YC, H * W * YC
        assert lib.yolo_conv_fwd(C.byref(d), st) == 0
        torch.cuda.synchronize()
        got = ybuf.float().cpu().numpy()
        assert (got[..., :y0] == -7.0).all() and (got[..., y0 + Cout:] == -7.0).all()
        np.testing.assert_array_equal(got[..., y0:y0 + Cout].transpose(0, 3, 1, 2), ref)


STATS_CASES = [(2, 32, 40, 70, 64, 3, 1), (3, 32, 33, 50, 64, 3, 2), (2, 64, 21, 37, 32, 1, 1), (2, 64, 13, 13, 128, 3, 1), (3, 64, 26, 26, 64, 1, 1), (2, 32, 16, 24, 24, 3, 1), (1, 128, 52, 52, 256, 3, 1),
               (4, 256, 13, 13, 512, 1, 1), (2, 64, 26, 26, 128, 3, 2), (5, 32, 7, 9, 16, 1, 1), (2, 128, 19, 19, 72, 3, 1), (2, 128, 26, 26, 256, 1, 1)]


STATS_ALGOS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 36, 37, 38, 39]


@pytest.mark.parametrize('case,algo,mode', [pytest.param(c, a, m, id='%s-a%d-m%d' % ('x'.join(map(str, c)), a, m))
                                            for c in STATS_CASES for a in STATS_ALGOS for m in (1, 2)
                                            if conv_variant(c, 'bf16', a, stats_mode=m) is not None])
def test_conv_statistics_epilogue(lib, cuda, case, algo, mode):
    """yolo_conv_desc.stats: Gluon BatchNorm's batch sums taken in the convolution's epilogue (per pixel-tile partial rows,
    summed in double by yolo_bn_train_*_partials).  Mode 1: sum(y), sum(y^2) of the stored bf16 output; mode 2 (a data
    gradient with accumulation into an existing gradient): sum(da), sum(da * xhat) of the BatchNorm behind the output.
    Checked through the whole call: the BatchNorm forward / backward fed by the partial rows must equal the one that
    reduces the stored tensor itself."""
    import ctypes as C
    N, Cin, H, W, Cout, k, stride = case
    rng = np.random.default_rng(7)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    xd = to_nhwc(x, 'bf16', cuda)
    wp = torch.empty(lib.yolo_packed_weight_bytes(Cout, Cin, k, L.BF16), dtype=torch.uint8, device=cuda)
    L.check(lib.yolo_pack_conv_weights(torch.from_numpy(w).to(cuda).data_ptr(), wp.data_ptr(), Cout, Cin, k, L.BF16, st), 'pack')
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    npix = N * Ho * Wo
    cp = lib.yolo_padded_channels(Cout)
    gamma = torch.from_numpy(rng.uniform(.5, 1.5, Cout).astype(np.float32)).to(cuda)
    beta = torch.from_numpy((0.2 * rng.standard_normal(Cout)).astype(np.float32)).to(cuda)
    y = torch.full((N, Ho, Wo, Cout), float('nan'), dtype=torch.bfloat16, device=cuda)
    d = L.ConvDesc()
    d.x, d.w_packed, d.y = xd.data_ptr(), wp.data_ptr(), y.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = N, H, W, Cin, Cout, k, stride, L.BF16, 1.0, algo
    ws = [torch.zeros(2 * Cout, dtype=torch.float64, device=cuda) for _ in range(4)]
    mk = lambda: (torch.empty(Cout, device=cuda), torch.empty(Cout, device=cuda))
    if mode == 1:
        d.stats, d.stats_mode = 1, 1
        rows = lib.yolo_conv_stats_rows(C.byref(d))
        assert rows > 0
        part = torch.full((rows, 2, cp), float('nan'), device=cuda)
        d.stats = part.data_ptr()
        assert lib.yolo_conv_fwd(C.byref(d), st) == 0
        (m1, i1), (m2, i2) = mk(), mk()
        z1, z2 = torch.empty_like(y), torch.empty_like(y)
        rm, rv = torch.zeros(Cout, device=cuda), torch.ones(Cout, device=cuda)
        assert lib.yolo_bn_train_fwd_partials(part.data_ptr(), rows, cp, y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None,
                                              z1.data_ptr(), m1.data_ptr(), i1.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                              ws[0].data_ptr(), ws[1].data_ptr(), 2 * Cout, npix, Cout, 1e-5, 0.9, 0.1, L.BF16, st) == 0
        assert lib.yolo_bn_train_fwd_pp(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, z2.data_ptr(), m2.data_ptr(),
                                        i2.data_ptr(), rm.data_ptr(), rv.data_ptr(), ws[2].data_ptr(), ws[3].data_ptr(), 2 * Cout,
                                        npix, Cout, 1e-5, 0.9, 0.1, L.BF16, st) == 0
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y.float()).all())
        np.testing.assert_allclose(m1.cpu().numpy(), m2.cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(i1.cpu().numpy(), i2.cpu().numpy(), rtol=1e-5)
        assert float((z1.float() - z2.float()).abs().max()) <= 1e-2 * float(z2.float().abs().max())
        return
    # mode 2: the convolution is a data gradient accumulating into an existing gradient; yb = the forward raw output of the
    # layer behind it, with that layer's saved statistics
    yb = torch.from_numpy(rng.standard_normal((N, Ho, Wo, Cout)).astype(np.float32)).to(cuda).bfloat16()
    acc0 = torch.from_numpy((0.5 * rng.standard_normal((N, Ho, Wo, Cout))).astype(np.float32)).to(cuda).bfloat16()
    mean = yb.float().mean(dim=(0, 1, 2)).contiguous()
    invstd = (1.0 / torch.sqrt(yb.float().var(dim=(0, 1, 2), unbiased=False) + 1e-5)).contiguous()
    y.copy_(acc0)
    d.residual = y.data_ptr()
    d.stats, d.stats_mode, d.stats_y = 1, 2, yb.data_ptr()
    d.stats_mean, d.stats_invstd, d.stats_gamma, d.stats_beta, d.stats_slope = (mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                                                               beta.data_ptr(), 0.1)
    rows = lib.yolo_conv_stats_rows(C.byref(d))
    assert rows > 0
    part = torch.full((rows, 2, cp), float('nan'), device=cuda)
    d.stats = part.data_ptr()
    assert lib.yolo_conv_fwd(C.byref(d), st) == 0
    dy1, dy2 = torch.empty_like(y), torch.empty_like(y)
    (g1, b1), (g2, b2) = mk(), mk()
    assert lib.yolo_bn_train_bwd_partials(part.data_ptr(), rows, cp, y.data_ptr(), yb.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                          gamma.data_ptr(), beta.data_ptr(), dy1.data_ptr(), g1.data_ptr(), b1.data_ptr(),
                                          ws[0].data_ptr(), ws[1].data_ptr(), 2 * Cout, npix, Cout, 0.1, L.BF16, st) == 0
    assert lib.yolo_bn_train_bwd_pp(y.data_ptr(), yb.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                    beta.data_ptr(), dy2.data_ptr(), g2.data_ptr(), b2.data_ptr(), ws[2].data_ptr(),
                                    ws[3].data_ptr(), 2 * Cout, npix, Cout, 0.1, L.BF16, st) == 0
    torch.cuda.synchronize()
    sc = float(g2.abs().max()) + float(b2.abs().max())
    np.testing.assert_allclose(g1.cpu().numpy(), g2.cpu().numpy(), rtol=1e-4, atol=1e-5 * sc)
    np.testing.assert_allclose(b1.cpu().numpy(), b2.cpu().numpy(), rtol=1e-4, atol=1e-5 * sc)
    assert float((dy1.float() - dy2.float()).abs().max()) <= 1e-2 * float(dy2.float().abs().max())


@pytest.mark.parametrize('shape', [(2, 64, 96, 32), (1, 37, 61, 32), (3, 20, 130, 64), (2, 416, 416, 32), (1, 9, 7, 16)])
def test_stem_statistics(lib, cuda, shape):
    """yolo_stem_conv_fwd_stats: the stem kernel's own batch sums (partial rows per wave) give the same BatchNorm as the
    reduction over the stored map; the stored map itself is bit-identical to yolo_stem_conv_fwd's."""
    N, H, W, Cout = shape
    rng = np.random.default_rng(3)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.from_numpy(rng.random((N, 3, H, W), dtype=np.float32)).to(cuda)
    w = torch.from_numpy((rng.standard_normal((Cout, 3, 3, 3)) / 5).astype(np.float32)).to(cuda)
    ones, zeros = torch.ones(Cout, device=cuda), torch.zeros(Cout, device=cuda)
    y1 = torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device=cuda); y2 = torch.empty_like(y1)
    rows = lib.yolo_stem_stats_rows(N, H, W, Cout)
    assert rows > 0
    part = torch.full((rows, 2, Cout), float('nan'), device=cuda)
    assert lib.yolo_stem_conv_fwd_stats(x.data_ptr(), w.data_ptr(), ones.data_ptr(), zeros.data_ptr(), y1.data_ptr(), N, H, W, 3,
                                        Cout, L.BF16, 1.0, part.data_ptr(), st) == 0
    assert lib.yolo_stem_conv_fwd(x.data_ptr(), w.data_ptr(), ones.data_ptr(), zeros.data_ptr(), y2.data_ptr(), N, H, W, 3, Cout,
                                  L.BF16, 1.0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y1.view(torch.int16), y2.view(torch.int16)) and bool(torch.isfinite(part).all())
    gamma, beta = torch.rand(Cout, device=cuda) + .5, torch.randn(Cout, device=cuda) * .1
    ws = [torch.zeros(2 * Cout, dtype=torch.float64, device=cuda) for _ in range(4)]
    m1, i1, m2, i2 = (torch.empty(Cout, device=cuda) for _ in range(4))
    z1, z2 = torch.empty_like(y1), torch.empty_like(y1)
    rm, rv = torch.zeros(Cout, device=cuda), torch.ones(Cout, device=cuda)
    npix = N * H * W
    assert lib.yolo_bn_train_fwd_partials(part.data_ptr(), rows, Cout, y1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None,
                                          z1.data_ptr(), m1.data_ptr(), i1.data_ptr(), rm.data_ptr(), rv.data_ptr(), ws[0].data_ptr(),
                                          ws[1].data_ptr(), 2 * Cout, npix, Cout, 1e-5, 0.9, 0.1, L.BF16, st) == 0
    assert lib.yolo_bn_train_fwd_pp(y1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, z2.data_ptr(), m2.data_ptr(),
                                    i2.data_ptr(), rm.data_ptr(), rv.data_ptr(), ws[2].data_ptr(), ws[3].data_ptr(), 2 * Cout, npix,
                                    Cout, 1e-5, 0.9, 0.1, L.BF16, st) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(m1.cpu().numpy(), m2.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(i1.cpu().numpy(), i2.cpu().numpy(), rtol=1e-5)
    assert float((z1.float() - z2.float()).abs().max()) <= 1e-2 * float(z2.float().abs().max())


# ---- fused tail 1x1 (yolo_conv_desc.tail_*): a 3x3 convolution + the 1x1 that follows it in ONE kernel ---------------------
TAIL_CASES = [
    # (N, Cin, H, W, Cout, stride, residual, tail_cout, tail_f32, strided y)
    (2, 128, 26, 26, 256, 1, True, 128, False, False),      # a stage-2 residual block's 3x3 + the next block's 1x1
    (3, 128, 19, 31, 256, 1, False, 128, False, True),      # main output into a channel slice of a wider buffer
    (2, 64, 40, 24, 128, 1, True, 64, False, False),        # Cout = 128: a 128-cout tile, or a 256-cout one whose upper couts are padding
    (3, 64, 52, 36, 128, 2, False, 64, False, False),       # stage 1's down-sampling conv + its first block's 1x1 (128-cout stride-2 tiles)
    (5, 128, 13, 13, 256, 1, False, 90, True, False),       # tip + YOLOOutput: fp32 logits, Cout 90, strided rows
    (2, 128, 52, 52, 256, 2, False, 128, False, False),     # the stage's down-sampling conv + the first block's 1x1
    (1, 256, 9, 70, 224, 1, True, 96, False, False),        # ragged counts: 224 = 7 chunks, 96 couts
    (33, 128, 8, 8, 256, 1, True, 128, False, False),       # tiles that cross image boundaries
]


@pytest.mark.parametrize('case,algo', [(c, a) for c in TAIL_CASES for a in [0] + ([10, 16, 18] + ([9, 17] if c[4] <= 128 else []) if c[5] == 2 else [2, 6] + ([7] if c[4] <= 128 else []))])
def test_conv_tail_1x1_fused_is_bit_identical(lib, cuda, case, algo):
    """yolo_conv_desc.tail_*: the 1x1 convolution behind a 3x3 one computed by the same kernel from the output tile it has just
    stored.  Against the two separate launches on the same buffers: the main output and the tail output must be bit-identical
    (same operands -- the stored bf16 outputs --, same K order), and the tail against torch on the rounded operands."""
    import ctypes as C
    from util import LDT, TDT
    N, Cin, H, W, Cout, stride, with_res, tcout, tf32, strided = case
    rng = np.random.default_rng(31)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    w1 = (rng.standard_normal((tcout, Cout, 1, 1)) / np.sqrt(Cout)).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    dt = L.BF16
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    xd = to_nhwc(x, 'bf16', cuda)
    r = rng.standard_normal((N, Cout, Ho, Wo)).astype(np.float32) if with_res else None
    rd = to_nhwc(r, 'bf16', cuda) if with_res else None

    def packed(wt, co, ci, k):
        wp = torch.empty(lib.yolo_packed_weight_bytes(co, ci, k, dt), dtype=torch.uint8, device=cuda)
        L.check(lib.yolo_pack_conv_weights(torch.from_numpy(wt).to(cuda).data_ptr(), wp.data_ptr(), co, ci, k, dt, st), 'pack')
        return wp

    def sb(co, seed):
        g = np.random.default_rng(seed)
        cp = lib.yolo_padded_channels(co)
        s_, b_ = torch.zeros(cp, device=cuda), torch.zeros(cp, device=cuda)
        s_[:co] = torch.from_numpy(g.uniform(.5, 1.5, co).astype(np.float32)).to(cuda)
        b_[:co] = torch.from_numpy((0.1 * g.standard_normal(co)).astype(np.float32)).to(cuda)
        return s_, b_

    wp, wp1 = packed(w, Cout, Cin, 3), packed(w1, tcout, Cout, 1)
    (sc, bi), (sc1, bi1) = sb(Cout, 1), sb(tcout, 2)
    ych = Cout + 64 if strided else Cout                      # (strided: y is the upper channel slice of a wider buffer)
    outs = []
    for fused in (False, True):
        ybuf = torch.full((N, Ho, Wo, ych), float('nan'), dtype=torch.bfloat16, device=cuda)
        y = ybuf[..., ych - Cout:]
        tpitch = tcout + 6 if tf32 else tcout                 # (fp32 logits: rows of a wider merged buffer)
        z = torch.full((N, Ho, Wo, tpitch), float('nan'), dtype=torch.float32 if tf32 else torch.bfloat16, device=cuda)
        d = L.ConvDesc()
        d.x, d.w_packed, d.scale, d.bias, d.y = xd.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), y.data_ptr()
        d.residual = rd.data_ptr() if with_res else None
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope, d.algo = N, H, W, Cin, Cout, 3, stride, dt, 0.1, algo
        d.y_pixel_stride, d.y_batch_stride = ych, Ho * Wo * ych
        if fused:
            d.tail_w_packed, d.tail_scale, d.tail_bias, d.tail_y = wp1.data_ptr(), sc1.data_ptr(), bi1.data_ptr(), z.data_ptr()
            d.tail_cout, d.tail_out_f32, d.tail_slope = tcout, int(tf32), (1.0 if tf32 else 0.1)
            d.tail_y_pixel_stride, d.tail_y_batch_stride = tpitch, Ho * Wo * tpitch
            assert lib.yolo_conv_fwd(C.byref(d), st) == 0
        else:
            # the same tile variant as the fused call (kernel families differ in their K order, i.e. in the last bit)
            for cand in ([algo] if algo else (([17, 9] if Cout <= 128 else []) + [18, 16, 10] if stride == 2 else ([7] if Cout <= 128 else []) + [6, 2])):
                d.algo = cand
                if lib.yolo_conv_fwd(C.byref(d), st) == 0:
                    break
            else:
                raise AssertionError('no 256-cout variant takes the plain convolution')
            d1 = L.ConvDesc()
            d1.x, d1.w_packed, d1.scale, d1.bias, d1.y = y.data_ptr(), wp1.data_ptr(), sc1.data_ptr(), bi1.data_ptr(), z.data_ptr()
            d1.N, d1.H, d1.W, d1.Cin, d1.Cout, d1.ksize, d1.stride, d1.dtype = N, Ho, Wo, Cout, tcout, 1, 1, dt
            d1.out_f32, d1.slope = int(tf32), (1.0 if tf32 else 0.1)
            d1.x_pixel_stride = ych
            d1.y_pixel_stride, d1.y_batch_stride = tpitch, Ho * Wo * tpitch
            assert lib.yolo_conv_fwd(C.byref(d1), st) == 0
        torch.cuda.synchronize()
        outs.append((y.contiguous().clone(), z[..., :tcout].contiguous().clone()))
    (y0, z0), (y1, z1) = outs
    assert not torch.isnan(y1.float()).any() and not torch.isnan(z1.float()).any()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))                     # the main output is unchanged ...
    it = torch.int32 if tf32 else torch.int16
    assert torch.equal(z0.view(it), z1.view(it))                                       # ... and the tail is the separate launch's, bit for bit
    # the tail against torch on the rounded operands
    yin = y1.float().permute(0, 3, 1, 2).cpu()
    ref = torch.nn.functional.conv2d(yin, torch.from_numpy(w1).to(torch.bfloat16).float())
    ref = ref * sc1[:tcout].cpu().view(1, -1, 1, 1) + bi1[:tcout].cpu().view(1, -1, 1, 1)
    if not tf32:
        ref = torch.where(ref > 0, ref, ref * 0.1).to(torch.bfloat16).float()
    got = z1.float().permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1.6e-2, atol=2e-2)


def test_conv_tail_rejects(lib, cuda):
    import ctypes as C
    p = torch.zeros(1 << 16, device=cuda).data_ptr()
    st = torch.cuda.current_stream().cuda_stream

    def desc(**kw):
        d = L.ConvDesc()
        d.x = d.w_packed = d.y = d.tail_w_packed = d.tail_y = p
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope = 1, 8, 8, 64, 256, 3, 1, L.BF16, 0.1
        d.tail_cout, d.tail_slope = 128, 0.1
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    assert lib.yolo_conv_fwd(C.byref(desc(Cout=512)), st) == -2               # two cout tiles: no block holds a pixel's channels
    assert lib.yolo_conv_fwd(C.byref(desc(tail_cout=256)), st) == -2
    assert lib.yolo_conv_fwd(C.byref(desc(ksize=1)), st) == -2
    assert lib.yolo_conv_fwd(C.byref(desc(dtype=L.F32)), st) == -2
    assert lib.yolo_conv_fwd(C.byref(desc(algo=8)), st) == -2                 # a 4-wave tile variant
    assert lib.yolo_conv_fwd(C.byref(desc(algo=7)), st) == -2                 # a 128-cout tile for 256 couts
    assert lib.yolo_conv_fwd(C.byref(desc(tail_y=None)), st) == -1
    assert lib.yolo_conv_fwd(C.byref(desc(tail_slope=2.0)), st) == -1
    assert lib.yolo_conv_fwd(C.byref(desc(upsample2x=1)), st) == -2
