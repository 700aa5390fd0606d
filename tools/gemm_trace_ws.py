#!/usr/bin/env python
"""Per-tile timeline of the wave-specialised split GEMM (needs a -DGN_SPLIT_TRACE=1 variant selected with GN_LIB_PATH)."""
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
assert lib.gn_debug_trace_ws(buf) == 0
t = torch.tensor(list(buf), dtype=torch.float64).reshape(64, 16, 8)
for wg in (0, 9, 63):
    print(f"workgroup {wg} (ticks): consumer K loop / acc->LDS + E barrier / whole tile ; producer tile start lag")
    for ti in range(1, 8):
        r, nxt = t[wg, ti], t[wg, ti + 1]
        print(f"  tile {ti}: K loop={int(r[1] - r[0]):6d}  stage+E={int(r[2] - r[1]):5d}  total={int(nxt[0] - r[0]):6d}  "
              f"producer start - consumer start={int(r[4] - r[0]):6d} | slab 3: consumer arrives {int(r[3] - r[0]):6d} leaves "
              f"{int(r[5] - r[0]):6d}; producer arrives {int(r[6] - r[0]):6d} leaves {int(r[7] - r[0]):6d}")
print(f"mean tile period {float((t[:, 2:8, 0] - t[:, 1:7, 0]).mean()):.0f} ticks, K loop {float((t[:, 1:8, 1] - t[:, 1:8, 0]).mean()):.0f}")
