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
    # the fused (two-launch) forward does not work in place: every block re-reads y at pixel 0 while another block writes z there
    q = p + 4096
    chk('bn_train_fwd_pp z == y', lib.yolo_bn_train_fwd_pp(p, p, p, None, p, p, p, p, p, q, q + 4096, 16, C.c_longlong(64), 8, f(1e-5), f(0.9), f(0.1), L.BF16, st))
    chk('conv_wgrad k=5', lib.yolo_conv_wgrad(p, p, p, 1, 8, 8, 8, 32, 5, 1, 0, L.BF16, p, st))
    chk('conv_wgrad N=0', lib.yolo_conv_wgrad(p, p, p, 0, 8, 8, 8, 32, 3, 1, 0, L.BF16, p, st))
    chk('conv_wgrad_algo algo=77', lib.yolo_conv_wgrad_algo(p, p, p, 1, 8, 8, 64, 64, 3, 1, 0, L.BF16, p, 77, st))
    chk('adam n=0', lib.yolo_adam_step(p, p, p, p, C.c_longlong(0), 1, f(1e-3), f(0.9), f(0.999), f(1e-8), f(1.0), st))
    chk('adam t=0', lib.yolo_adam_step(p, p, p, p, C.c_longlong(8), 0, f(1e-3), f(0.9), f(0.999), f(1e-8), f(1.0), st))
    chk('adam_dev NULL slot', lib.yolo_adam_step_dev(p, p, p, p, C.c_longlong(8), 1, f(1e-3), f(0.9), f(0.999), f(1e-8), None, st))
    chk('add n=0', lib.yolo_add(p, p, p, C.c_longlong(0), L.BF16, st))
    chk('dilate2x dy larger than the target allows', lib.yolo_dilate2x(p, p, 1, 8, 8, 5, 5, 8, L.BF16, st))
    chk('dilate2x Ho=0', lib.yolo_dilate2x(p, p, 1, 8, 8, 0, 4, 8, L.BF16, st))

    # second sweep: the fused / elementwise / training families with NULL pointers and empty extents
    g2 = L.GridDesc()
    s5 = (C.c_float * 5)(0.1, 0.01, 10.0, 0.0, 0.3)
    chk('stem_conv_fwd NULL', lib.yolo_stem_conv_fwd(None, p, p, p, p, 1, 8, 8, 3, 32, L.BF16, f(0.1), st))
    chk('stem_conv_fwd N=0', lib.yolo_stem_conv_fwd(p, p, p, p, p, 0, 8, 8, 3, 32, L.BF16, f(0.1), st))
    chk('stem_conv_fwd cin=5', lib.yolo_stem_conv_fwd(p, p, p, p, p, 1, 8, 8, 5, 32, L.BF16, f(0.1), st))
    chk('stem_down_fwd NULL', lib.yolo_stem_down_fwd(None, p, p, p, p, p, p, p, 1, 8, 8, 32, 64, L.BF16, f(0.1), st))
    chk('stem_down_fwd widths 16 -> 64', lib.yolo_stem_down_fwd(p, p, p, p, p, p, p, p, 1, 8, 8, 16, 64, L.BF16, f(0.1), st))
    chk('res_block_fwd NULL', lib.yolo_res_block_fwd(None, p, p, p, p, p, p, p, 1, 8, 8, 64, L.BF16, f(0.1), st))
    chk('res_block_fwd C=96', lib.yolo_res_block_fwd(p, p, p, p, p, p, p, p, 1, 8, 8, 96, L.BF16, f(0.1), st))
    chk('res_block_fwd f32', lib.yolo_res_block_fwd(p, p, p, p, p, p, p, p, 1, 8, 8, 64, L.F32, f(0.1), st))
    chk('composite n=0', lib.yolo_composite(p, p, p, p, C.c_longlong(0), st))
    chk('composite NULL', lib.yolo_composite(None, p, p, p, C.c_longlong(8), st))
    chk('composite_unit NULL', lib.yolo_composite_unit(p, None, p, p, C.c_longlong(8), st))
    chk('upsample2x_concat NULL', lib.yolo_upsample2x_concat(None, p, p, 1, 8, 8, 8, 8, L.BF16, st))
    chk('upsample2x_concat C not 16-byte', lib.yolo_upsample2x_concat(p, p, p, 1, 8, 8, 3, 8, L.BF16, st))
    chk('predict_lp hw=0', lib.yolo_predict_lp(p, p, p, 10, 0, 0, f(45), f(60), f(45), st))
    chk('predict_lp C=3', lib.yolo_predict_lp(p, p, p, 3, 4, 4, f(45), f(60), f(45), st))
    chk('image_u8_to_nchw NULL', lib.yolo_image_u8_to_nchw(None, p, 1, 8, 8, 3, st))
    chk('nhwc_to_nchw C=0', lib.yolo_nhwc_to_nchw(p, p, 1, 0, 8, 8, L.BF16, st))
    chk('pack_conv_weights_dgrad k=5', lib.yolo_pack_conv_weights_dgrad(p, p, 32, 32, 5, L.BF16, st))
    chk('pack_conv_weights_dgrad_s2 NULL', lib.yolo_pack_conv_weights_dgrad_s2(None, p, 32, 32, L.BF16, st))
    chk('conv_dgrad_s2 NULL desc', lib.yolo_conv_dgrad_s2(None, st))
    chk('pack_batch_blocks k=7', lib.yolo_pack_batch_blocks(32, 32, 7, L.BF16))
    chk('pack_conv_weights_batch n=0', lib.yolo_pack_conv_weights_batch(p, p, 0, C.c_longlong(4), L.BF16, st))
    chk('pack_pair_blocks Cin=24', lib.yolo_pack_pair_blocks(32, 24, 3))
    chk('pack_conv_weights_pairs NULL', lib.yolo_pack_conv_weights_pairs(None, p, 1, C.c_longlong(4), st))
    chk('bias_grad C=0', lib.yolo_bias_grad(p, p, C.c_longlong(8), 0, C.c_longlong(8), L.BF16, st))
    chk('gather_rows NULL', lib.yolo_gather_rows(None, p, 8, C.c_longlong(8), 8, 8, C.c_longlong(8), C.c_longlong(8), L.BF16, st))
    chk('upsample2x_concat_bwd NULL', lib.yolo_upsample2x_concat_bwd(None, p, p, 1, 8, 8, 8, 8, 0, 0, L.BF16, st))
    chk('assign_targets NULL grid', lib.yolo_assign_targets(p, p, p, 1, 1, 4, None, st))
    chk('assign_targets B=0', lib.yolo_assign_targets(p, p, p, 0, 1, 4, C.byref(g2), st))
    chk('loss_fwd_bwd B=0', lib.yolo_loss_fwd_bwd(p, p, p, p, 0, 10, 10, 1, s5, f(1.0), f(0.1), st))
    chk('loss_fwd_bwd NULL scales', lib.yolo_loss_fwd_bwd(p, p, p, p, 1, 10, 10, 1, None, f(1.0), f(0.1), st))
    chk('assign_targets_lp B=0', lib.yolo_assign_targets_lp(p, p, 0, 1, 4, 8, 8, 16, 64, f(45), f(60), f(45), st))
    chk('loss_lp_fwd_bwd NULL', lib.yolo_loss_lp_fwd_bwd(None, p, p, p, 1, 10, 10, 1, s5, f(1.0), f(0.1), st))
    chk('conv_wgrad_workspace_bytes k=5', lib.yolo_conv_wgrad_workspace_bytes(32, 32, 5, L.BF16))
    chk('nms_workspace_bytes B=0', lib.yolo_nms_workspace_bytes(0, 10, 30, 1, 400))
    chk('nms_select_workspace_bytes B=0', lib.yolo_nms_select_workspace_bytes(0))
    chk('bn_train_fwd_partials rows=0', lib.yolo_bn_train_fwd_partials(p, 0, 32, p, p, p, None, p, p, p, p, p, p, p, 16, C.c_longlong(64), 8, f(1e-5), f(0.9), f(0.1), L.BF16, st))
    chk('bn_train_bwd_partials f32', lib.yolo_bn_train_bwd_partials(p, 4, 32, p, p, p, p, p, p, p, p, p, p, p, 16, C.c_longlong(64), 8, f(0.1), L.F32, st))
    torch.cuda.synchronize()
    assert not bad, bad
    assert float(buf.abs().sum()) == 0                     # nothing was launched on the scratch buffer
