import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from yolo_amd import lib as L
exec(open('tools/erratum/pk_bisect.py').read().split("def main():")[0])
dev = torch.device('cuda:0')
ship, pk = L.load(), load(PK)
spin = C.CDLL('tools/_build/libpk_spin.so')
vp = C.c_void_p
g = torch.Generator(device='cpu').manual_seed(3)
spin_out = torch.zeros(16, device=dev)
for (N, H, W, Cc) in ((4, 208, 208, 64), (64, 52, 52, 256)):
    y = torch.randn((N, H, W, Cc), generator=g).to(dev).to(torch.bfloat16)
    dz = (0.01 * torch.randn((N, H, W, Cc), generator=g)).to(dev).to(torch.bfloat16)
    yf = y.float(); mean = yf.mean(dim=(0, 1, 2)).contiguous(); invstd = (1.0 / torch.sqrt(yf.var(dim=(0, 1, 2), unbiased=False) + 1e-5)).contiguous()
    gamma = (0.5 + torch.rand(Cc, generator=g)).to(dev); beta = (0.1 * torch.randn(Cc, generator=g)).to(dev)
    npix = N * H * W
    ws = [torch.zeros(4096, dtype=torch.float64, device=dev) for _ in range(2)]
    dgam, dbet = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    def bn(lib, out, k):
        ws[k & 1].zero_()
        assert lib.yolo_bn_train_bwd_pp(dz.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), ws[k & 1].data_ptr(), ws[(k & 1) ^ 1].data_ptr(), 4096, npix, Cc, 0.1, L.BF16, torch.cuda.current_stream().cuda_stream) == 0
    ref = torch.empty_like(y); bn(ship, ref, 0); torch.cuda.synchronize()
    for name, lib in (('packed', pk), ('shipped', ship)):
        for bits, bname in ((0x7fc00000, 'NaN'), (0, 'zero'), (0x3f800000, '1.0')):
            bad = nan = zero = ev = 0
            for r in range(10):
                out = torch.full_like(y, 3.0)
                torch.cuda.synchronize()
                spin.spin_poison(C.c_uint(bits), vp(spin_out.data_ptr()), vp(torch.cuda.current_stream().cuda_stream))
                bn(lib, out, r)
                torch.cuda.synchronize()
                ne = out.view(torch.int16) != ref.view(torch.int16)
                k = int(ne.sum())
                if k:
                    ev += 1; bad += k; nan += int(torch.isnan(out[ne].float()).sum()); zero += int((out[ne] == 0).sum())
            print('%dx%dx%dx%d %-7s BatchNorm backward ALONE after a register-file poison of %-4s: rounds with a mismatch %2d / 10, elements %7d (NaN %d, zeros %d)' % (N, H, W, Cc, name, bname, ev, bad, nan, zero), flush=True)
