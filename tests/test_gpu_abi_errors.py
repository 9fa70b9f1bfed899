"""Error behaviour of the C ABI (include/yolo_amd.h: 0 ok, -1 invalid argument, -2 unsupported shape): every entry point called
with NULL pointers, zero / negative sizes, unsupported enums or inconsistent shapes RETURNS a negative status -- it neither
launches nor crashes."""
import ctypes as C

import pytest
import torch

from yolo_amd import lib as L

pytestmark = pytest.mark.gpu


def test_invalid_arguments_return_a_status(cuda):
    lib = L.load()
    buf = torch.zeros(1 << 20, device=cuda)
    p = buf.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    f = C.c_float

    def chk(name, rc):
        if not (rc == -1 or rc == -2):
            bad.append((name, rc))
    bad = []
    f = C.c_float
    # conv
    d = L.ConvDesc()
    chk('conv_fwd NULL desc', lib.yolo_conv_fwd(None, st))
    chk('conv_fwd zeroed desc', lib.yolo_conv_fwd(C.byref(d), st))
    def desc(**kw):
        d = L.ConvDesc()
        d.x = d.w_packed = d.y = p
        d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.dtype, d.slope = 1, 8, 8, 8, 32, 3, 1, L.BF16, 0.1
        for k, v in kw.items(): setattr(d, k, v)
        return d
    for name, kw in (('N=0', dict(N=0)), ('H=-1', dict(H=-1)), ('Cin=7 (not 16-byte)', dict(Cin=7)), ('ksize=5', dict(ksize=5)), ('stride=3', dict(stride=3)),
                     ('dtype=9', dict(dtype=9)), ('slope=2', dict(slope=2.0)), ('slope=-1', dict(slope=-1.0)), ('algo=99', dict(algo=99)), ('Cout=0', dict(Cout=0)),
                     ('stats without mode', dict(stats=p, stats_mode=7))):
        dd = desc(**kw)
        chk('conv_fwd ' + name, lib.yolo_conv_fwd(C.byref(dd), st))
    chk('packed_weight_bytes k=5', lib.yolo_packed_weight_bytes(32, 32, 5, L.BF16))
    chk('packed_weight_bytes dtype=9', lib.yolo_packed_weight_bytes(32, 32, 3, 9))
    chk('pack_conv_weights NULL', lib.yolo_pack_conv_weights(None, p, 32, 32, 3, L.BF16, st))
    chk('pack_conv_weights Cout=0', lib.yolo_pack_conv_weights(p, p, 0, 32, 3, L.BF16, st))
    chk('fold_bn C=0', lib.yolo_fold_bn(p, p, p, p, f(1e-5), p, p, 0, st))
    chk('fold_bn gamma without beta', lib.yolo_fold_bn(p, None, p, p, f(1e-5), p, p, 8, st))
    chk('fold_bn NULL outputs', lib.yolo_fold_bn(p, p, p, p, f(1e-5), None, p, 8, st))
    chk('nchw_to_nhwc C=0', lib.yolo_nchw_to_nhwc(p, p, 1, 0, 8, 8, 8, L.BF16, st))
    chk('nchw_to_nhwc Cpad<C', lib.yolo_nchw_to_nhwc(p, p, 1, 3, 8, 8, 2, L.BF16, st))
    g = L.GridDesc()
    if True:
        chk('decode NULL grid', lib.yolo_decode(p, p, 1, 30, None, st))
        chk('decode zeroed grid', lib.yolo_decode(p, p, 1, 30, C.byref(g), st))
        chk('decode B=0', lib.yolo_decode(p, p, 0, 30, C.byref(g), st))
        chk('predict_top1 C=5', lib.yolo_predict_top1(p, p, p, 1, 5, C.byref(g), st))
        chk('decode_scores mode=3', lib.yolo_decode_scores(p, p, p, 1, 30, C.byref(g), 3, st))
    chk('nms_scores C=5', lib.yolo_nms_scores(p, p, 1, 10, 5, 1, st))
    chk('nms_scores C=200', lib.yolo_nms_scores(p, p, 1, 10, 200, 1, st))
    chk('nms_from_scores topk=0', lib.yolo_nms_from_scores(p, p, 1, 10, 30, 24, f(0.01), f(0.45), 0, 100, p, p, p, None, st))
    chk('nms_from_scores topk=600', lib.yolo_nms_from_scores(p, p, 1, 10, 30, 24, f(0.01), f(0.45), 600, 100, p, p, p, None, st))
    chk('nms_from_scores post_nms=0', lib.yolo_nms_from_scores(p, p, 1, 10, 30, 24, f(0.01), f(0.45), 400, 0, p, p, p, None, st))
    chk('nms no workspace', lib.yolo_nms(p, 1, 10, 30, 1, f(0.01), f(0.45), 400, 100, p, p, p, None, st))
    chk('nms mode=5', lib.yolo_nms(p, 1, 10, 30, 5, f(0.01), f(0.45), 400, 100, p, p, p, p, st))
    chk('iou n=0', lib.yolo_iou_ltrb_vs_yxhw(p, p, p, 0, st))
    chk('iou NULL', lib.yolo_iou_ltrb_vs_cltrb(None, p, p, 4, st))
    chk('bn_train_fwd C=7', lib.yolo_bn_train_fwd(p, p, p, None, p, p, p, p, p, p, C.c_longlong(64), 7, f(1e-5), f(0.9), f(0.1), L.BF16, st))
    chk('bn_train_fwd npix=0', lib.yolo_bn_train_fwd(p, p, p, None, p, p, p, p, p, p, C.c_longlong(0), 8, f(1e-5), f(0.9), f(0.1), L.BF16, st))
    chk('bn_train_bwd dtype=9', lib.yolo_bn_train_bwd(p, p, p, p, p, p, p, p, p, p, C.c_longlong(64), 8, f(0.1), 9, st))
    chk('bn_train_bwd_pp ws == zero_next', lib.yolo_bn_train_bwd_pp(p, p, p, p, p, p, p, p, p, p, p, 16, C.c_longlong(64), 8, f(0.1), L.BF16, st))
    chk('conv_wgrad k=5', lib.yolo_conv_wgrad(p, p, p, 1, 8, 8, 8, 32, 5, 1, 0, L.BF16, p, st))
    chk('conv_wgrad N=0', lib.yolo_conv_wgrad(p, p, p, 0, 8, 8, 8, 32, 3, 1, 0, L.BF16, p, st))
    chk('conv_wgrad_algo algo=77', lib.yolo_conv_wgrad_algo(p, p, p, 1, 8, 8, 64, 64, 3, 1, 0, L.BF16, p, 77, st))
    chk('adam n=0', lib.yolo_adam_step(p, p, p, p, C.c_longlong(0), 1, f(1e-3), f(0.9), f(0.999), f(1e-8), f(1.0), st))
    chk('adam t=0', lib.yolo_adam_step(p, p, p, p, C.c_longlong(8), 0, f(1e-3), f(0.9), f(0.999), f(1e-8), f(1.0), st))
    chk('adam_dev NULL slot', lib.yolo_adam_step_dev(p, p, p, p, C.c_longlong(8), 1, f(1e-3), f(0.9), f(0.999), f(1e-8), None, st))
    chk('add n=0', lib.yolo_add(p, p, p, C.c_longlong(0), L.BF16, st))
    chk('dilate2x dy larger than the target allows', lib.yolo_dilate2x(p, p, 1, 8, 8, 5, 5, 8, L.BF16, st))
    chk('dilate2x Ho=0', lib.yolo_dilate2x(p, p, 1, 8, 8, 0, 4, 8, L.BF16, st))

    torch.cuda.synchronize()
    assert not bad, bad
    assert float(buf.abs().sum()) == 0                     # nothing was launched on the scratch buffer
