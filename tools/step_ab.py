#!/usr/bin/env python
"""In-process A/B of the energy+force step for a boolean attribute of GotenNet (fuse_eqff, ...):
python tools/step_ab.py ATTR [workload batch lmax]   -> ms/step with ATTR = True / False, alternating, 3 rounds."""
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

attr = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "rmd17_aspirin"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
lmax = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda")
torch.manual_seed(0)
rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                            num_heads=8, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
pos, batch, z = (v.to(dev) for v in synthetic.make_batch(workload, B, seed=0))
ei, ed, ev = distance(pos, batch, 5.0, 32)
mp = molecule_ptr(batch, B)
ef = EnergyForces(rep, head, check_edges=False)


def timed(n=20):
    for _ in range(3):
        ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        e, f = ef(z, ei, ed, ev, batch, B, mol_ptr=mp)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, e, f


res = {True: [], False: []}
out = {}
for r in range(3):
    for flag in (True, False):
        setattr(rep, attr, flag)
        ms, e, f = timed()
        res[flag].append(ms)
        out[flag] = (e.clone(), f.clone())
de = float((out[True][0] - out[False][0]).abs().max() / out[False][0].abs().max())
df = float((out[True][1] - out[False][1]).abs().max() / out[False][1].abs().max())
print(f"{workload} b={B} lmax={lmax} {attr}: True {min(res[True]):.3f} ms {['%.3f' % v for v in res[True]]} | "
      f"False {min(res[False]):.3f} ms {['%.3f' % v for v in res[False]]} | rel diff e {de:.1e} f {df:.1e}")
