import torch, time
dev = torch.device("cuda")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
N = 16384*256
for dtype in (torch.int16, torch.int32):
    keys = torch.randint(0, 4096, (N,), device=dev).to(dtype)
    keys[torch.rand(N, device=dev) > 0.25] = 32767   # 75 % inactive sentinel
    print(dtype, "sort 4.2M", round(bench(lambda: torch.sort(keys)), 3), "ms")
    act = keys[keys != 32767].contiguous()
    print(dtype, "sort 1.0M (active only)", act.numel(), round(bench(lambda: torch.sort(act)), 3), "ms")
ks = torch.sort(keys.to(torch.int32))[0]
q = torch.arange(4097, device=dev, dtype=torch.int32)
print("searchsorted", round(bench(lambda: torch.searchsorted(ks, q)), 3), "ms")
print("nonzero/compaction of 4.2M mask", round(bench(lambda: torch.nonzero(keys != 32767)), 3), "ms")
print("bincount 1M", round(bench(lambda: torch.bincount(act.to(torch.int64), minlength=4096)), 3), "ms")
