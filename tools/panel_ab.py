#!/usr/bin/env python
"""Projection launches at the atom-sized shapes of the step + the step itself, for the library GN_LIB_PATH selects:
   GN_LIB_PATH=gotennet_amd/variants/lib_nopanel.so python tools/panel_ab.py ; python tools/panel_ab.py
-> us per launch, max error against an fp64 product relative to max|C|, ms per energy+forces step (C2 batch, one molecule)."""
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
tag = os.environ.get("GN_LIB_PATH", "product")


def shape(M, N, K, it=50, epi=False):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / 16
    b = torch.randn(N, device=dev, generator=g)
    C = torch.empty(M, N, device=dev)
    kw = dict(mode="f16x2")
    res = gate = None
    if epi:
        res = torch.randn(M, N, device=dev, generator=g)
        gate = torch.randn(M, N, device=dev, generator=g)
        kw.update(act=(0, N), res=res, gate=gate)
    run = lambda: engine.gemm(A, K, W, b, C, N, M, N, K, **kw)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        run()
        with torch.cuda.graph(gr, stream=st):                 # host-free timing: `it` launches per replay
            for _ in range(it):
                run()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(4):
            gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    it *= 4
    ref = A.double() @ W.double().t() + b.double()
    if epi:
        ref = res.double() + torch.nn.functional.silu(ref) * gate.double()
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    print(f"[{tag}] gemm {M}x{N}x{K}{' +epi' if epi else ''}: {e0.elapsed_time(e1) * 1e3 / it:7.2f} us  err {err:.1e}")


for s in ((2688, 512, 256), (2688, 256, 512), (2688, 256, 256), (2688, 1280, 256), (2688, 768, 256), (21504, 256, 256),
          (21, 512, 256), (441, 1536, 256), (2688, 256, 128), (54368, 256, 256), (429, 256, 1536), (168, 256, 768),
          (2688, 256, 1536), (2688, 256, 768), (21504, 256, 768), (5376, 1280, 256)):
    shape(*s)
shape(2688, 256, 256, epi=True)
shape(54368, 256, 256, epi=True)


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
    for r in range(3):
        for _ in range(3):
            ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            e, f = ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / n)
    print(f"[{tag}] step {workload} b={B} lmax={lmax}: {best:.3f} ms  e[0] {float(e[0]):.6f} |f| {float(f.abs().sum()):.4f}")


step("rmd17_aspirin", 128)
step("rmd17_aspirin", 128, lmax=4, n=10)
step("rmd17_aspirin", 1, n=50)
step("rmd17_aspirin", 8, n=50)
