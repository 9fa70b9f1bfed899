import torch
dev = torch.device('cuda:0')
for mb in (64, 256, 1024, 2048):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    y = torch.empty_like(x)
    for name, f, traffic in (('copy', lambda: y.copy_(x), 2), ('read-sum', lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32), 1), ('fill', lambda: y.zero_(), 1), ('add_', lambda: y.add_(x), 3)):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        print('%5d MB %-9s %.2f TB/s' % (mb, name, traffic * n * 2 / t / 1e12), flush=True)
