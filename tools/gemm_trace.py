#!/usr/bin/env python
"""Per-tile phase timeline of the split GEMM (needs a -DGN_SPLIT_TRACE=1 variant selected with GN_LIB_PATH)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import _lib, engine  # noqa: E402

M, N, K = (int(os.environ.get(k, d)) for k, d in (("GM", 54368), ("GN", 1536), ("GK", 256)))
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16; C = torch.empty(M, N, device="cuda")
for _ in range(3):
    engine.gemm(A, K, W, None, C, N, M, N, K)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * (64 * 16 * 8))()
assert lib.gn_debug_trace(buf) == 0
t = torch.tensor(list(buf), dtype=torch.float64).reshape(64, 16, 8)
names = ["stash0+B0+barrier", "K loop", "acc->LDS+barrier", "stores issued", "final barrier", "next-tile setup"]
for wg in (0, 1, 8, 9, 63):
    print(f"workgroup {wg}: per-tile phase durations in cycles (tiles 1..8)")
    for ti in range(1, 9):
        r = t[wg, ti]
        nxt = t[wg, ti + 1, 0]
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], nxt - r[5]]
        print(f"  tile {ti}: " + "  ".join(f"{n}={int(v):6d}" for n, v in zip(names, d)) + f"   total={int(nxt - r[0]):6d}")
tot = (t[:, 2:9, 0] - t[:, 1:8, 0]).mean()
print(f"mean tile period over 64 workgroups: {float(tot):.0f} cycles; K loop share "
      f"{float((t[:, 1:8, 2] - t[:, 1:8, 1]).mean() / tot):.2f}, epilogue share {float((t[:, 1:8, 5] - t[:, 1:8, 2]).mean() / tot):.2f}")
