import torch, time
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): y.copy_(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"copy {mb} MB: {dt*1e6:.1f} us  {2*mb/1024/dt/1e3:.2f} TB/s (read+write)")
    a = torch.empty_like(x)
    for _ in range(3): torch.add(x, y, out=a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): torch.add(x, y, out=a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"add  {mb} MB: {dt*1e6:.1f} us  {3*mb/1024/dt/1e3:.2f} TB/s (2 reads + write)")
