import torch, time
n = 131072 * 2500
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
def t(fn, k=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
print("fill  TB/s", 4 * n / t(lambda: a.zero_()) / 1e12)
print("copy  TB/s (r+w)", 8 * n / t(lambda: b.copy_(a)) / 1e12)
print("sum   TB/s", 4 * n / t(lambda: a.sum()) / 1e12)
