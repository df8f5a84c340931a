"""GPU probe: device-to-device copy rate against buffer size (what the 256 MB last-level cache is worth)."""
import torch, time
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    reps = max(10, 20000 // mb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("copy %5d MB -> %5d MB: %.4f ms  %.2f TB/s (read+write)" % (mb, mb, ms, 2 * mb * 1.048576e6 / ms / 1e9), flush=True)
    del a, b
