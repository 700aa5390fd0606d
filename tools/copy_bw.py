"""Practical HBM streaming rate on this box: torch device-to-device copies and an out-of-place add of growing size (read + write bytes
over the event time, best of 5 x 10 launches).  The 8 TB/s the rooflines are priced against is the nameplate; working sets beyond the
256 MiB Infinity Cache stream slower than cache-resident ones."""
import torch
for mib in (64, 128, 256, 512, 1024, 2048, 4096):
    n = mib * 1024 * 1024 // 4
    a, b = torch.empty(n, device="cuda").normal_(), torch.empty(n, device="cuda")
    c = torch.empty(n, device="cuda").normal_()
    def timed(f):
        best = 1e9
        for _ in range(5):
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        return best
    t_copy = timed(lambda: b.copy_(a))
    t_add = timed(lambda: torch.add(a, c, out=b))
    print(f"{mib:5d} MiB per tensor: copy {2 * n * 4 / t_copy / 1e9:6.2f} TB/s ({t_copy * 1e3:7.1f} us), a + c -> b {3 * n * 4 / t_add / 1e9:6.2f} TB/s ({t_add * 1e3:7.1f} us)", flush=True)
