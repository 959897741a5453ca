import torch, time, numpy as np
n = 1936 * 1024 * 1024 // 8
d = torch.rand(n, dtype=torch.float64, device='cuda')
h = torch.empty(n, dtype=torch.float64)
hp = torch.empty(n, dtype=torch.float64).pin_memory()
for name, dst in (('pageable', h), ('pinned', hp)):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, ['%.1f ms' % (x * 1e3) for x in ts], '%.1f GB/s' % (n * 8 / min(ts) / 1e9))
a = np.empty(n)
t0 = time.perf_counter(); r = torch.cuda.cudart().cudaHostRegister(a.ctypes.data, a.nbytes, 0); print('register', r, '%.0f ms' % ((time.perf_counter() - t0) * 1e3))
ta = torch.from_numpy(a)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ta.copy_(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('registered numpy', ['%.1f ms' % (x * 1e3) for x in ts])
