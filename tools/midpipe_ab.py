#!/usr/bin/env python
"""Mid-size projection launches (N <= 256, K a multiple of 256) stand-alone through hipGraph replays, for the library
GN_LIB_PATH selects (A/B of gn_gemm_midpipe.hip against the slab / panel kernels):
   GN_LIB_PATH=gotennet_amd/variants/lib_nomid.so python tools/midpipe_ab.py ; python tools/midpipe_ab.py
-> us per launch, TFLOP/s, GB/s of compulsory traffic, max error against an fp64 product relative to max|C|."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import engine  # noqa: E402

dev = torch.device("cuda")
tag = os.path.basename(os.environ.get("GN_LIB_PATH", "product"))
E, N, D = 54368, 2688, 8


def timed(run, it=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        run()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(it):
                run()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * it)


def report(name, probs, nbytes, flops, check):
    run = lambda: engine.gemm_group(probs, mode="f16x2")
    us = timed(run)
    run()
    torch.cuda.synchronize()
    err = check()
    print(f"[{tag}] {name:44s} {us:8.2f} us  {flops / us / 1e6:7.1f} TF  {nbytes / us / 1e3:7.1f} GB/s  err {err:.1e}", flush=True)


def r_(g):
    return lambda *s: torch.randn(*s, device=dev, generator=g)


def edge_product(epi, rider=True, M=E):
    g = torch.Generator(device="cuda").manual_seed(7)
    r = r_(g)
    K = Nn = 256
    A, W, b, C = r(M, K), r(Nn, K) / 16, r(Nn), torch.empty(M, Nn, device=dev)
    prob = dict(A=A, lda=K, W=W, bias=b, C=C, ldc=Nn, rows=M, nout=Nn, K=K)
    nbytes = 2 * 4 * M * Nn
    res = gate = None
    if epi == "gate":                     # t' = t + SiLU(W_t t + b) * w, pre-activation kept (forward HTR update)
        gate = r(M, Nn)
        P = torch.empty(M, Nn, device=dev)
        prob.update(act=(0, Nn), res=A, gate=gate, pre_out=P)
        res = A
        nbytes = 4 * 4 * M * Nn
    elif epi == "res":                    # its input-gradient: gt + g_pre_t W_t
        res = r(M, Nn)
        prob.update(res=res, bias=None)
        nbytes = 3 * 4 * M * Nn
    probs = [prob]
    flops = 2.0 * M * Nn * K
    if rider:
        A2, W2, b2, C2 = r(N, 512), r(256, 512) / 16, r(256), torch.empty(N, 256, device=dev)
        probs.append(dict(A=A2, lda=512, W=W2, bias=b2, C=C2, ldc=256, rows=N, nout=256, K=512, act=(0, 256)))
        flops += 2.0 * N * 256 * 512

    def check():
        ref = A.double() @ W.double().t() + (b.double() if epi != "res" else 0)
        if epi == "gate":
            ref = res.double() + torch.nn.functional.silu(ref) * gate.double()
        elif epi == "res":
            ref = res.double() + ref
        e = float((C.double() - ref).abs().max() / ref.abs().max())
        if rider:
            ref2 = torch.nn.functional.silu(A2.double() @ W2.double().t() + b2.double())
            e = max(e, float((C2.double() - ref2).abs().max() / ref2.abs().max()))
        return e
    report(f"[{M}x256x256 {epi or 'plain'}{' + rider' if rider else ''}]", probs, nbytes, flops, check)


def x_products(lmax=2, n=N):
    """forward: X W_vu, X W_vq, X^l W_vk^l (row-mapped)"""
    g = torch.Generator(device="cuda").manual_seed(11)
    r = r_(g)
    Dd = (lmax + 1) ** 2 - 1
    X = r(n, Dd, 256)
    Ws = [r(256, 256) / 16 for _ in range(2 + lmax)]
    outs = [torch.zeros(n, Dd, 256, device=dev) for _ in range(3)]
    probs = [dict(A=X, lda=256, W=Ws[0], C=outs[0], ldc=256, rows=n * Dd, nout=256, K=256),
             dict(A=X, lda=256, W=Ws[1], C=outs[1], ldc=256, rows=n * Dd, nout=256, K=256)]
    off = 0
    for l in range(1, lmax + 1):
        cnt = 2 * l + 1
        probs.append(dict(A=X, lda=256, W=Ws[1 + l], C=outs[2], ldc=256, rows=n * cnt, nout=256, K=256, rowmap=(cnt, Dd, off)))
        off += cnt

    def check():
        e = 0.0
        for o, w in ((outs[0], Ws[0]), (outs[1], Ws[1])):
            ref = X.double() @ w.double().t()
            e = max(e, float((o.double() - ref).abs().max() / ref.abs().max()))
        off = 0
        for l in range(1, lmax + 1):
            cnt = 2 * l + 1
            ref = X[:, off:off + cnt].double() @ Ws[1 + l].double().t()
            e = max(e, float((outs[2][:, off:off + cnt].double() - ref).abs().max() / ref.abs().max()))
            off += cnt
        return e
    rows = n * Dd
    for i0 in range(0, len(probs), 4):
        chunk = probs[i0:i0 + 4]
        rr = sum(q["rows"] for q in chunk)
        report(f"[X products lmax {lmax} #{i0 // 4}: {'+'.join(str(q['rows']) for q in chunk)} x256x256]", chunk,
               4 * rows * 256 + 4 * rr * 256, 2.0 * rr * 256 * 256, check if i0 + 4 >= len(probs) else (lambda: float("nan")))


def x_cat(lmax=2, n=N):
    """backward: gX1 = gX + [gXp | gEQ | gEK] [W_vu^T | W_vq^T | W_vk_l^T] per degree block (K-segmented A)"""
    g = torch.Generator(device="cuda").manual_seed(13)
    r = r_(g)
    Dd = (lmax + 1) ** 2 - 1
    X0, X1, X2, Rr = r(n, Dd, 256), r(n, Dd, 256) * 8, r(n, Dd, 256) / 8, r(n, Dd, 256)
    Cc = torch.zeros(n, Dd, 256, device=dev)
    Wc = [r(256, 768) / 16 for _ in range(lmax)]
    probs, off = [], 0
    for l in range(1, lmax + 1):
        cnt = 2 * l + 1
        probs.append(dict(A=X0, A2=X1, A3=X2, a_seg=256, lda=256, W=Wc[l - 1], C=Cc, ldc=256, rows=n * cnt, nout=256, K=768,
                          rowmap=(cnt, Dd, off), res=Rr))
        off += cnt

    def check():
        e, off = 0.0, 0
        for l in range(1, lmax + 1):
            cnt = 2 * l + 1
            s = slice(off, off + cnt)
            w = Wc[l - 1].double()
            ref = Rr[:, s].double() + X0[:, s].double() @ w[:, :256].t() + X1[:, s].double() @ w[:, 256:512].t() + X2[:, s].double() @ w[:, 512:].t()
            e = max(e, float((Cc[:, s].double() - ref).abs().max() / ref.abs().max()))
            off += cnt
        return e
    rows = n * Dd
    report(f"[X-cat lmax {lmax}: {'+'.join(str(q['rows']) for q in probs)} x256x768 res]", probs, 5 * 4 * rows * 256,
           2.0 * rows * 256 * 768, check)


if __name__ == "__main__":
    edge_product("gate")
    edge_product("gate", rider=False)
    edge_product("res")
    edge_product(None, rider=False)
    x_products(2)
    x_cat(2)
    x_products(4)
    x_cat(4)
    edge_product("gate", M=82000)
    edge_product("gate", M=120000)
