import torch, time
dev = torch.device("cuda")
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
n = 1 << 30  # 4 GiB of floats
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
ms = timeit(lambda: a.fill_(1.0)); print(f"fill  {4*n/ms/1e6:.0f} GB/s")
ms = timeit(lambda: b.copy_(a)); print(f"copy  {8*n/ms/1e6:.0f} GB/s (r+w)")
ms = timeit(lambda: a.sum()); print(f"sum   {4*n/ms/1e6:.0f} GB/s")
ms = timeit(lambda: torch.add(a, b, out=b)); print(f"add   {12*n/ms/1e6:.0f} GB/s (2r+w)")
