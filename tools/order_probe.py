#!/usr/bin/env python
"""Does the atom numbering matter?  Energy+force step of a workload with the atoms of every molecule in input (random) order
vs renumbered along a Morton curve (graph.spatial_order).  python tools/order_probe.py [workload batch lmax cell]"""
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

workload = sys.argv[1] if len(sys.argv) > 1 else "md22_nanotube"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lmax = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cell = float(sys.argv[4]) if len(sys.argv) > 4 else 2.5
dev = torch.device("cuda")
torch.manual_seed(0)
rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                            num_heads=8, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
pos, batch, z = (v.to(dev) for v in synthetic.make_batch(workload, B, seed=0))


def morton(pos, batch, cell):
    lo = torch.zeros((int(batch.max()) + 1, 3), device=pos.device).index_reduce_(0, batch, pos, "amin", include_self=False)
    q = ((pos - lo[batch]) / cell).floor().to(torch.int64).clamp_(0, 1023)
    code = torch.zeros_like(q[:, 0])
    for b in range(10):
        for d in range(3):
            code |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return torch.argsort(batch * (1 << 30) + code, stable=True)


def run(pos, z):
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    mp = molecule_ptr(batch, B)
    ef = EnergyForces(rep, head, check_edges=False)
    for _ in range(3):
        ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        e, f = ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / 20, e, f, ei.shape[1]


perm = morton(pos, batch, cell)
for r in range(2):
    t0, e0, f0, E0 = run(pos, z)
    t1, e1, f1, E1 = run(pos[perm], z[perm])
    f1o = torch.empty_like(f1)
    f1o[perm] = f1
    print(f"{workload} b={B} lmax={lmax}: input order {t0:.3f} ms (E={E0}) | Morton (cell {cell}) {t1:.3f} ms (E={E1}) | "
          f"energy diff {float((e1 - e0).abs().max() / e0.abs().max()):.1e}, force diff {float((f1o - f0).abs().max() / f0.abs().max()):.1e}")
