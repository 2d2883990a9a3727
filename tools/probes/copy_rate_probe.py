"""Device-to-device copy rate (bytes read + bytes written per second) at a few sizes: what a kernel that reads and writes in equal parts
can expect from HBM (and from the Infinity Cache while the two buffers fit in it).  Run on the GPU box."""
import torch
for mb in (16, 66, 132, 264, 1024, 4096):
    n = mb * 1000 * 1000
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    for _ in range(5):
        b.copy_(a)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(10, 20000 // mb)
    s.record()
    for _ in range(reps):
        b.copy_(a)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print("copy of %5d MB: %8.1f us  = %6.0f GB/s read + write" % (mb, us, 2 * n / us / 1e3))
