#!/usr/bin/env python
"""Host time to ENQUEUE one fused energy+force step (no synchronisation) against its wall time on the GPU, C2 batch: is the
eager path ever waiting for Python?  (MI355X, round 3: 3.7 ms of enqueue per 7.7 ms step -- the host runs 2x ahead.)"""
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import gotennet_amd
from gotennet_amd import synthetic
from gotennet_amd.graph import distance
from gotennet_amd.outputs import Atomwise, molecule_ptr
from gotennet_amd.pipeline import EnergyForces
dev = "cuda"
torch.manual_seed(0)
rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0), num_heads=8, scale_edge=False, lmax=2, sep_dir=True, sep_tensor=True).to(dev).eval()
head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
pos, batch, z = synthetic.make_batch("rmd17_aspirin", 128, seed=0)
pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)
ei, ed, ev = distance(pos, batch, 5.0, 32)
mp = molecule_ptr(batch, 128)
ef = EnergyForces(rep, head, check_edges=False)
for _ in range(5): ef(z, ei, ed, ev, batch, 128, mol_ptr=mp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): ef(z, ei, ed, ev, batch, 128, mol_ptr=mp)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"python enqueue {1e3*(t1-t0)/20:.3f} ms/step, wall {1e3*(t2-t0)/20:.3f} ms/step")
