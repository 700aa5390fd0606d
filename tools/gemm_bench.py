#!/usr/bin/env python
"""Micro-benchmark + check of gn_gemm_ex on the shapes the C2 step issues (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import engine  # noqa: E402

SHAPES = [  # (M, N, K, mode)
    (54368, 1536, 256, "plain"), (54368, 256, 1536, "dsilu"), (54368, 256, 256, "gate"), (54368, 256, 256, "dsilu_gate"),
    (21504, 256, 256, "plain"), (13440, 256, 256, "rowmap"), (2688, 1280, 256, "silu_pro"), (2688, 256, 1280, "plain"),
    (2688, 1024, 256, "plain"), (2688, 256, 1024, "dsilu"), (2688, 512, 256, "plain"), (2688, 256, 512, "plain"),
    (2688, 256, 256, "plain"), (54368, 512, 32, "plain"), (54368, 32, 512, "plain"),
]
torch.manual_seed(0)
dev = "cuda"
for (M, N, K, mode) in SHAPES:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    kw = {}
    ref = None
    if mode == "plain":
        ref = A @ W.T + b
    elif mode == "silu_pro":
        kw = dict(pro=(1, 0, K)); ref = torch.nn.functional.silu(A) @ W.T + b
    elif mode in ("dsilu", "dsilu_gate"):
        P = torch.randn(M, K, device=dev)
        s = torch.sigmoid(P); d = s * (1 + P * (1 - s))
        kw = dict(pro=(2, 0, K), a_pre=P, ldp=K)
        Aeff = A * d
        if mode == "dsilu_gate":
            G = torch.randn(M, K, device=dev); kw.update(a_gate=G, ldg=K); Aeff = Aeff * G
        R = torch.randn(M, N, device=dev); kw.update(res=R)
        ref = R + Aeff @ W.T + b
    elif mode == "gate":
        R = torch.randn(M, N, device=dev); G = torch.randn(M, N, device=dev)
        kw = dict(act=(0, N), res=R, gate=G); ref = R + torch.nn.functional.silu(A @ W.T + b) * G
    elif mode == "rowmap":
        kw = dict(rowmap=(5, 8, 3)); ref = None
    if mode == "rowmap":
        A = torch.randn((M // 5) * 8, K, device=dev); C = torch.zeros((M // 5) * 8, N, device=dev)
        run = lambda: engine.gemm(A, K, W, b, C, N, M, N, K, **kw)
        run(); torch.cuda.synchronize()
        Av = A.view(M // 5, 8, K)[:, 3:8].reshape(M, K)
        ref = Av @ W.T + b
        got = C.view(M // 5, 8, N)[:, 3:8].reshape(M, N)
    else:
        run = lambda: engine.gemm(A, K, W, b, C, N, M, N, K, **kw)
        run(); torch.cuda.synchronize()
        got = C
    err = float((got - ref).abs().max() / ref.abs().max())
    # fp64 check of the plain case
    if mode == 'plain':
        r64 = A.double() @ W.double().T + b.double()
        err = float((got.double() - r64).abs().max() / r64.abs().max())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{M:6d}x{N:5d}x{K:5d} {mode:11s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF  relerr {err:.1e}")
