"""Experiment: what the hardware asks for the memory work of one C5 generation --
10 M random 2-byte gathers over 2.6 GB, and a 1.8 GB stream -- with torch's own
kernels, to set beside k_step's time."""
import torch
dev = torch.device("cuda", 0)
n = 10_000_000
a = torch.zeros(1_300_000_000, dtype=torch.int16, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
idx = torch.randint(0, a.numel() - 64, (n,), device=dev, generator=g)
idx_sorted = idx.sort().values
src = torch.zeros(1_180_000_000 // 8, dtype=torch.float64, device=dev)
dst = torch.empty(600_000_000 // 8, dtype=torch.float64, device=dev)


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print(f"10M random 2-byte gathers over 2.6 GB : {timed(lambda: a[idx]):8.1f} us")
print(f"the same, indices sorted              : {timed(lambda: a[idx_sorted]):8.1f} us")
i4 = torch.stack([idx, idx + 1, idx + 8, idx + 9], 1).reshape(-1)
print(f"4 nodes of a cell per gather (40M)    : {timed(lambda: a[i4]):8.1f} us")
print(f"read 1.18 GB (sum)                    : {timed(lambda: src.sum()):8.1f} us")
print(f"write 0.6 GB (fill)                   : {timed(lambda: dst.fill_(1.0)):8.1f} us")
print(f"copy 0.6 GB -> 0.6 GB                 : {timed(lambda: dst.copy_(src[:dst.numel()])):8.1f} us")
