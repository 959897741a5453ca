import torch, time, os, sys, subprocess
n = 1900 * 1024 * 1024 // 8
x = torch.empty(n, dtype=torch.float64, device='cuda'); y = torch.rand(n, dtype=torch.float64, device='cuda')
for name, fn, nbytes in (('fill (write only)', lambda: x.fill_(1.0), n * 8), ('copy (read + write)', lambda: x.copy_(y), 2 * n * 8), ('sum (read only)', lambda: y.sum(), n * 8)):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print('%-20s %.3f ms  %.2f TB/s' % (name, min(ts), nbytes / min(ts) / 1e9))
