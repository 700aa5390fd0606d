#!/usr/bin/env python
"""The projection shapes that carry the C2 step, in the current GEMM mode, with the epilogues the step uses (GPU box).
GN_LIB_PATH selects a tuning variant built by tools/variants.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import engine  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
E, N, F, M = 54368, 2688, 256, 5
r = lambda *s: torch.randn(*s, device=dev)


def timed(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


def check(C, ref):
    return float((C.double() - ref).abs().max() / ref.abs().max())


t, h = r(E, F), r(N, F)
We, be, Wn1, bn1 = r(6 * F, F) / 16, r(6 * F), r(4 * F, F) / 16, r(4 * F)
eproj, nact = torch.empty(E, 6 * F, device=dev), torch.empty(N, 4 * F, device=dev)
fwd = lambda: engine.gemm_group([dict(A=t, lda=F, W=We, bias=be, C=eproj, ldc=6 * F, rows=E, nout=6 * F, K=F),
                                 dict(A=h, lda=F, W=Wn1, bias=bn1, C=nact, ldc=4 * F, rows=N, nout=4 * F, K=F, act=(2 * F, 4 * F))])
us = timed(fwd)
err = check(eproj[:4096], t[:4096].double() @ We.double().T + be.double())
print(f"fwd eproj  [E x 1536 x 256 + N x 1024 x 256]     {us:8.1f} us  {2 * (E * 6 * F * F + N * 4 * F * F) / us / 1e6:7.1f} TF  err {err:.1e}")

g_e, WeT, gt_in, gt_b = r(E, 6 * F), r(F, 6 * F) / 40, r(E, F), torch.empty(E, F, device=dev)
g_x, WsT, g_np, pre = r(N, M * F), r(F, M * F) / 36, torch.empty(N, 4 * F, device=dev), r(N, 4 * F)
bwd = lambda: engine.gemm_group([dict(A=g_e, lda=6 * F, W=WeT, C=gt_b, ldc=F, rows=E, nout=F, K=6 * F, res=gt_in),
                                 dict(A=g_x, lda=M * F, W=WsT, C=g_np, ldc=4 * F, rows=N, nout=F, K=M * F, c_off=2 * F, dgate=pre, g_off=2 * F),
                                 dict(A=g_x, lda=M * F, W=WsT, C=g_np, ldc=4 * F, rows=N, nout=F, K=M * F, c_off=3 * F, dgate=pre, g_off=3 * F)])
us = timed(bwd)
err = check(gt_b[:4096], gt_in[:4096].double() + g_e[:4096].double() @ WeT.double().T)
print(f"bwd W_e^T  [E x 256 x 1536 + 2 (N x 256 x 1280)] {us:8.1f} us  {2 * (E * 6 * F * F + 2 * N * M * F * F) / us / 1e6:7.1f} TF  err {err:.1e}")

Wt, bt, w, t2 = r(F, F) / 16, r(F), r(E, F), torch.empty(E, F, device=dev)
ctx, Wm0, g1 = r(N, 2 * F), r(F, 2 * F) / 22, torch.empty(N, F, device=dev)
gam = lambda: engine.gemm_group([dict(A=t, lda=F, W=Wt, bias=bt, C=t2, ldc=F, rows=E, nout=F, K=F, act=(0, F), res=t, gate=w),
                                 dict(A=ctx, lda=2 * F, W=Wm0, C=g1, ldc=F, rows=N, nout=F, K=2 * F, act=(0, F))])
us = timed(gam)
ref = t[:4096].double() + torch.nn.functional.silu(t[:4096].double() @ Wt.double().T + bt.double()) * w[:4096].double()
print(f"gamma_t    [E x 256 x 256 gated + N x 256 x 512] {us:8.1f} us  {2 * (E * F * F + N * 2 * F * F) / us / 1e6:7.1f} TF  err {check(t2[:4096], ref):.1e}")

D = 8
X, Wv, Xp, EQ = r(N * D, F), r(F, F) / 16, torch.empty(N * D, F, device=dev), torch.empty(N * D, F, device=dev)
EK = torch.zeros(N * D, F, device=dev)
xp = lambda: engine.gemm_group([dict(A=X, lda=F, W=Wv, C=Xp, ldc=F, rows=N * D, nout=F, K=F),
                                dict(A=X, lda=F, W=Wv, C=EQ, ldc=F, rows=N * D, nout=F, K=F),
                                dict(A=X, lda=F, W=Wv, C=EK, ldc=F, rows=N * 3, nout=F, K=F, rowmap=(3, D, 0)),
                                dict(A=X, lda=F, W=Wv, C=EK, ldc=F, rows=N * 5, nout=F, K=F, rowmap=(5, D, 3))])
us = timed(xp)
print(f"X products [2 (ND x 256 x 256) + 3N.. + 5N..]    {us:8.1f} us  {2 * (3 * N * D * F * F) / us / 1e6:7.1f} TF  err {check(Xp[:4096], X[:4096].double() @ Wv.double().T):.1e}")

xs, Ws2, nact2 = torch.empty(N, M * F, device=dev), r(M * F, F) / 16, r(N, 4 * F)
xv = lambda: engine.gemm_group([dict(A=nact2, lda=4 * F, W=Ws2, C=xs, ldc=M * F, rows=N, nout=M * F, K=F, a_off=2 * F),
                                dict(A=nact2, lda=4 * F, W=Ws2, C=xs, ldc=M * F, rows=N, nout=M * F, K=F, a_off=3 * F)])
us = timed(xv)
print(f"x / v      [2 (N x 1280 x 256)]                   {us:8.1f} us  {2 * (2 * N * M * F * F) / us / 1e6:7.1f} TF")
