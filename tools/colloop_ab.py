#!/usr/bin/env python
"""Edge-sized projection launches (stand-alone, host-free timing through hipGraph replays) + the energy+forces step, for
the library GN_LIB_PATH selects (A/B of the column-loop kernel, gn_gemm_colloop.hip):
   GN_LIB_PATH=gotennet_amd/variants/lib_nocl.so python tools/colloop_ab.py ; python tools/colloop_ab.py
-> us per launch, TFLOP/s, max error against an fp64 product relative to max|C|; ms per step (C2 batch; lmax 2 and 4)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gotennet_amd  # noqa: E402
from gotennet_amd import engine, synthetic  # noqa: E402
from gotennet_amd.graph import distance  # noqa: E402
from gotennet_amd.outputs import Atomwise, molecule_ptr  # noqa: E402
from gotennet_amd.pipeline import EnergyForces  # noqa: E402

dev = torch.device("cuda")
tag = os.path.basename(os.environ.get("GN_LIB_PATH", "product"))
E, N = 54368, 2688


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


def shape(M, Nn, K, epi=None, rider=None, check=True):
    g = torch.Generator(device="cuda").manual_seed(M + Nn + K)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    A, W, b, C = r(M, K), r(Nn, K) / 16, r(Nn), torch.empty(M, Nn, device=dev)
    prob = dict(A=A, lda=K, W=W, bias=b, C=C, ldc=Nn, rows=M, nout=Nn, K=K)
    res = gate = None
    if epi == "gate":
        res, gate = r(M, Nn), r(M, Nn)
        prob.update(act=(0, Nn), res=res, gate=gate, pre_out=torch.empty(M, Nn, device=dev))
    elif epi == "res":
        res = r(M, Nn)
        prob.update(res=res)
    probs = [prob]
    flops = 2.0 * M * Nn * K
    if rider is not None:
        for (m2, n2, k2) in rider:
            probs.append(dict(A=r(m2, k2), lda=k2, W=r(n2, k2) / 16, C=torch.empty(m2, n2, device=dev), ldc=n2, rows=m2, nout=n2, K=k2))
            flops += 2.0 * m2 * n2 * k2
    run = lambda: engine.gemm_group(probs, mode="f16x2")
    us = timed(run)
    err = float("nan")
    if check:
        ref = A.double() @ W.double().t() + b.double()
        if epi == "gate":
            ref = res.double() + torch.nn.functional.silu(ref) * gate.double()
        elif epi == "res":
            ref = res.double() + ref
        err = float((C.double() - ref).abs().max() / ref.abs().max())
    print(f"[{tag}] gemm {M}x{Nn}x{K} {epi or 'plain':5s}{' +riders' if rider else ''}: {us:8.2f} us  {flops / us / 1e6:7.1f} TF  err {err:.1e}", flush=True)


def step(workload, B, lmax=2, n=20):
    torch.manual_seed(0)
    rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
    head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
    pos, batch, z = (v.to(dev) for v in synthetic.make_batch(workload, B, seed=0))
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    mp = molecule_ptr(batch, B)
    ef = EnergyForces(rep, head, check_edges=False)
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            e, f = ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / n)
    print(f"[{tag}] step {workload} b={B} lmax={lmax}: {best:.3f} ms  e[0] {float(e[0]):.6f} |f| {float(f.abs().sum()):.4f}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["shapes", "steps"]
    if "shapes" in what:
        shape(E, 1536, 256, rider=[(N, 1024, 256)])
        shape(E, 1536, 256)
        shape(E, 1024, 256)
        shape(E, 2560, 256)
        shape(E, 256, 1536, epi="res", rider=[(N, 256, 1280), (N, 256, 1280), (N, 256, 512)])
        shape(E, 256, 1536, epi="res")
        shape(E, 256, 256, epi="gate", rider=[(N, 256, 512)])
        shape(E, 256, 256, epi="gate")
        shape(E, 256, 256, epi="res")
        shape(E, 256, 256)
        shape(21504, 256, 256)
        shape(21504, 256, 768, epi="res")
        shape(E, 512, 32)
    if "steps" in what:
        step("rmd17_aspirin", 128)
        step("rmd17_aspirin", 128, lmax=4, n=10)
        step("md22_ac_ala3", 64, n=10)
        step("md22_nanotube", 8, lmax=3, n=10)
