"""Does a second read of a buffer that fits the memory-side cache (256 MB) run faster than a read from HBM?
Reads of buffers of several sizes, each repeated back to back, timed by events: GB/s by size (needs a GPU)."""
import torch

dev = torch.device("cuda", 0)
big = torch.ones(300_000_000, dtype=torch.float64, device=dev)          # 2.4 GB
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2400):
    n = mb * 1_000_000 // 8
    x = big[:n]
    reps = max(4, 4000 // mb)
    for _ in range(3):
        x.sum()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        x.sum()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{mb:5d} MB read again and again: {ms * 1e3:8.1f} us  {n * 8 / ms / 1e9:7.2f} TB/s", flush=True)
# copy (read + write) of the same sizes
for mb in (32, 64, 96, 128, 256, 1024):
    n = mb * 1_000_000 // 8
    x, y = big[:n], big[150_000_000:150_000_000 + n]
    reps = max(4, 4000 // mb)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{mb:5d} MB copied again and again: {ms * 1e3:8.1f} us  {2 * n * 8 / ms / 1e9:7.2f} TB/s (read + write)", flush=True)
