#!/usr/bin/env python
"""Per-workgroup residency of one row-stationary GEMM launch (GN_RS_ABL=128 build switch): start / end in s_memrealtime
ticks (100 MHz), CU / SE / XCC from HW_ID -- are two workgroups co-resident per CU?"""
import ctypes
import os
import sys

os.environ["GN_RS_ABL"] = os.environ.get("GN_RS_ABL", "128")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import _lib, engine  # noqa: E402

M, N, K = 54368, int(os.environ.get("GN", 1536)), 256
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
C = torch.empty(M, N, device="cuda")
for _ in range(3):
    engine.gemm(A, K, W, b, C, N, M, N, K)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 4096)()
lib.gn_debug_rs_trace.argtypes = [ctypes.c_void_p]
assert lib.gn_debug_rs_trace(buf) == 0
rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(1024) if buf[4 * i + 1]]
t0 = min(r[0] for r in rows)
print(f"{len(rows)} workgroups; launch span {(max(r[1] for r in rows) - t0) / 100:.1f} us")
durs = sorted((r[1] - r[0]) / 100 for r in rows)
print(f"workgroup duration us: min {durs[0]:.1f} median {durs[len(durs) // 2]:.1f} max {durs[-1]:.1f}")
starts = sorted((r[0] - t0) / 100 for r in rows)
print("start times us (deciles):", " ".join(f"{starts[int(q * (len(starts) - 1) / 10)]:.1f}" for q in range(11)))
# HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... (gfx9 layout)
from collections import Counter
cu = Counter((r[2] >> 8) & 0xffff for r in rows)
print("workgroups per (cu, sh, se, ...) key: histogram of counts:", sorted(Counter(cu.values()).items()))
# concurrency: how many workgroups are alive at the median time
mid = t0 + (max(r[1] for r in rows) - t0) // 4
print("alive at 25% of the span:", sum(1 for r in rows if r[0] <= mid < r[1]))
mid = t0 + (max(r[1] for r in rows) - t0) * 3 // 4
print("alive at 75% of the span:", sum(1 for r in rows if r[0] <= mid < r[1]))
if int(os.environ["GN_RS_ABL"]) & 256:
    pb = (ctypes.c_longlong * (64 * 2 * 8 * 8))()
    lib.gn_debug_rs_phases.argtypes = [ctypes.c_void_p]
    assert lib.gn_debug_rs_phases(pb) == 0
    import statistics
    names = ["wait(vmcnt)", "barrier", "epi/A/bias/succ/dma-issue", "mfma phase", "tail->next top"]
    acc = [[] for _ in names]
    for blk in range(64):
        for w in range(2):
            for tl in range(7):
                b0 = ((blk * 2 + w) * 8 + tl) * 8
                st = [pb[b0 + k] for k in range(5)]
                nxt = pb[b0 + 8]
                if not all(st) or not nxt:
                    continue
                for k in range(4):
                    acc[k].append(st[k + 1] - st[k])
                acc[4].append(nxt - st[4])
    for n, v in zip(names, acc):
        if v:
            v.sort()
            print(f"{n:28s} median {v[len(v) // 2]:7d}  p10 {v[len(v) // 10]:7d}  p90 {v[9 * len(v) // 10]:7d}  (shader cycles, n={len(v)})")
    tot = sum(statistics.median(v) for v in acc if v)
    print(f"sum of medians per tile: {tot:.0f} cycles")
