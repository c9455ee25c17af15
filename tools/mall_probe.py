"""Does a write -> read of a buffer hit the Infinity Cache (MALL)?  Times a device copy dst <- src
right after src was written, for buffer sizes around the 256 MB cache."""
import torch

dev = torch.device("cuda", 0)
for mb in (16, 32, 64, 128, 192, 256, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    best_w = best_r = 1e9
    for _ in range(20):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        a.fill_(1.0)          # write mb
        e1.record()
        s = a.sum()           # read mb right after
        e2.record()
        torch.cuda.synchronize()
        best_w = min(best_w, e0.elapsed_time(e1))
        best_r = min(best_r, e1.elapsed_time(e2))
    print(f"{mb:5d} MB  write {mb / 1024 / (best_w * 1e-3) / 1e3:6.2f} TB/s   read-after-write {mb / 1024 / (best_r * 1e-3) / 1e3:6.2f} TB/s")
