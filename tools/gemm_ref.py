#!/usr/bin/env python
"""Calibration only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt, bf16) sustains on plain
GEMMs of the same M x N x K as the long-K convolutions' implicit GEMMs -- no im2col, no epilogue, operands contiguous.
    python tools/gemm_ref.py"""
import torch
dev = torch.device('cuda:0')
# (label, M = N*Ho*Wo, N = Cout, K = 9*Cin)
SHAPES = [('608 bs64 19^2 1024->2048', 64 * 19 * 19, 2048, 9 * 1024), ('608 bs64 38^2 512->1024', 64 * 38 * 38, 1024, 9 * 512),
          ('608 bs64 76^2 256->512', 64 * 76 * 76, 512, 9 * 256), ('416 bs32 13^2 1024->2048', 32 * 13 * 13, 2048, 9 * 1024),
          ('416 bs32 26^2 512->1024', 32 * 26 * 26, 1024, 9 * 512), ('416 bs32 52^2 256->512', 32 * 52 * 52, 512, 9 * 256),
          ('416 bs32 52^2 128->256', 32 * 52 * 52, 256, 9 * 128), ('416 bs32 26^2 1x1 1024->512', 32 * 26 * 26, 512, 1024),
          ('square 8192', 8192, 8192, 8192)]
for label, M, N, K in SHAPES:
    a = torch.randn((M, K), device=dev).bfloat16()
    b = torch.randn((N, K), device=dev).bfloat16()
    best = 1e9
    for bt in (b, b.t().contiguous().t()):          # weights K-major / N-major
        f = (lambda: a @ bt.t()) if bt is b else (lambda: a @ bt.t())
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    print('%-32s M %7d N %5d K %5d  %8.1f us  %7.1f TFLOP/s' % (label, M, N, K, best, 2.0 * M * N * K / best / 1e6), flush=True)
