#!/usr/bin/env python
"""Row-stationary K=256 GEMM (gn_gemm_rs.hip): time of the path's shapes; run under GN_RS_ABL=<bits> / GN_RS_GRID=<n> /
GN_GEMM_RS=0 for the ablations (wrong results with GN_RS_ABL != 0: timing only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import engine  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
tag = " ".join(f"{k}={os.environ[k]}" for k in ("GN_GEMM_RS", "GN_RS_ABL", "GN_RS_GRID", "GN_GEMM_RS_MIN_TILES") if k in os.environ) or "default"
out = []
for (M, N, mode) in [(54368, 1536, "plain"), (54368, 256, "gate"), (21504, 256, "plain"), (2688, 1280, "plain"), (2688, 512, "plain")]:
    K = 256
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 16; b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    kw = {}
    if mode == "gate":
        kw = dict(act=(0, N), res=torch.randn(M, N, device=dev), gate=torch.randn(M, N, device=dev))
    run = lambda: engine.gemm(A, K, W, b, C, N, M, N, K, **kw)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{M}x{N}:{e0.elapsed_time(e1) / 20 * 1e3:7.1f}")
print(f"{tag:28s} " + "  ".join(out))
