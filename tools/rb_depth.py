import os, sys, ctypes as C
sys.path.insert(0, '/root/repo')
import torch
from yolo_amd import lib as L
from yolo_amd.net import CarNet
from yolo_amd.spec import darknet53_spec
dev = torch.device('cuda:0')
net = CarNet(darknet53_spec(), device=dev)
lib, st = net._lib, L.stream_ptr()
for (n, hw, c) in ((32, 208, 64), (32, 104, 128), (64, 304, 64), (64, 152, 128)):
    x = torch.randn((n, hw, hw, c), device=dev).bfloat16()
    y = torch.empty_like(x)
    w1 = torch.randn((c // 2, c, 1, 1), device=dev) * 0.05; w2 = torch.randn((c, c // 2, 3, 3), device=dev) * 0.05
    wp1 = torch.empty(lib.yolo_packed_weight_bytes(c // 2, c, 1, 1), dtype=torch.uint8, device=dev)
    wp2 = torch.empty(lib.yolo_packed_weight_bytes(c, c // 2, 3, 1), dtype=torch.uint8, device=dev)
    lib.yolo_pack_conv_weights(w1.data_ptr(), wp1.data_ptr(), c // 2, c, 1, 1, st); lib.yolo_pack_conv_weights(w2.data_ptr(), wp2.data_ptr(), c, c // 2, 3, 1, st)
    s1 = torch.ones(lib.yolo_padded_channels(c // 2), device=dev); b1 = torch.zeros_like(s1)
    s2 = torch.ones(lib.yolo_padded_channels(c), device=dev); b2 = torch.zeros_like(s2)
    f = lambda: lib.yolo_res_block_fwd(x.data_ptr(), wp1.data_ptr(), s1.data_ptr(), b1.data_ptr(), wp2.data_ptr(), s2.data_ptr(), b2.data_ptr(), y.data_ptr(), n, hw, hw, c, 1, 0.1, st)
    assert f() == 0
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); e1.synchronize()
    print('D', os.environ.get('YOLO_RB_D', '3'), n, hw, c, '%.1f us' % (e0.elapsed_time(e1) / 30 * 1e3), flush=True)
