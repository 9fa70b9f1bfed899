"""Time yolo_res_block_fwd on one shape (ablation bits via YOLO_RB_AB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L
lib = L.load()
dev = torch.device('cuda:0')
N, H, W, C = [int(v) for v in sys.argv[1:5]]
st = torch.cuda.current_stream().cuda_stream
x = torch.randn((N, H, W, C), device=dev).bfloat16()
y = torch.empty_like(x)
def pack(co, ci, k):
    w = torch.randn((co, ci, k, k), device=dev) * 0.05
    wp = torch.empty(lib.yolo_packed_weight_bytes(co, ci, k, L.BF16), dtype=torch.uint8, device=dev)
    L.check(lib.yolo_pack_conv_weights(w.data_ptr(), wp.data_ptr(), co, ci, k, L.BF16, st), 'pack')
    return wp
wp1, wp2 = pack(C // 2, C, 1), pack(C, C // 2, 3)
s = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
def run():
    L.check(lib.yolo_res_block_fwd(x.data_ptr(), wp1.data_ptr(), s.data_ptr(), b.data_ptr(), wp2.data_ptr(), s.data_ptr(), b.data_ptr(),
                                   y.data_ptr(), N, H, W, C, L.BF16, 0.1, st), 'rb')
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print('ab=%s  %dx%dx%dx%d  %.1f us  (%.2f TB/s of read+write once)' % (os.environ.get('YOLO_RB_AB', '0'), N, H, W, C, us, 2 * x.numel() * 2 / us / 1e6))
