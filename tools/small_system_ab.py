#!/usr/bin/env python
"""Small systems as one hipGraph replay per step (EnergyForces(replay=True)): ms/step with GotenNet.fuse_eqff forced on / off for 1 / 8 / 32 molecules.   python tools/small_system_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gotennet_amd  # noqa: E402
from gotennet_amd import synthetic  # noqa: E402
from gotennet_amd.graph import distance  # noqa: E402
from gotennet_amd.outputs import Atomwise, molecule_ptr  # noqa: E402
from gotennet_amd.pipeline import EnergyForces  # noqa: E402

dev = torch.device("cuda")


def run(B, forces, n=100, **attrs):
    torch.manual_seed(0)
    rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=False, lmax=2, sep_dir=True, sep_tensor=True).to(dev).eval()
    for k, v in attrs.items():
        setattr(rep, k, v)
    head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
    pos, batch, z = (v.to(dev) for v in synthetic.make_batch("rmd17_aspirin", B, seed=0))
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    mp = molecule_ptr(batch, B)
    if forces:
        ef = EnergyForces(rep, head, check_edges=False, replay=True)
        step = lambda: ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
    else:                                            # energy only: capture the eager forward ourselves
        ef = EnergyForces(rep, head, check_edges=False)
        for _ in range(3):
            ef(z, ei, ed, ev, batch, B, mol_ptr=mp, forces=False)
        torch.cuda.synchronize()
        st, gr = torch.cuda.Stream(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            ef(z, ei, ed, ev, batch, B, mol_ptr=mp, forces=False)
            with torch.cuda.graph(gr, stream=st):
                out = ef(z, ei, ed, ev, batch, B, mol_ptr=mp, forces=False)
        torch.cuda.synchronize()
        step = gr.replay
    best = 1e9
    for r in range(3):
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / n)
    return best


for B in (1, 8, 32):
    a, b = run(B, True, fuse_eqff=True), run(B, True, fuse_eqff=False)
    print(f"b={B} energy+forces, one replay per step: fuse_eqff on {a:.3f} ms | off {b:.3f} ms")
