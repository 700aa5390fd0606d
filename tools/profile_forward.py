#!/usr/bin/env python
"""Per-kernel time breakdown of one forward at a BASELINE config (GPU box)."""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gotennet_amd  # noqa: E402
from gotennet_amd import _lib, engine, synthetic  # noqa: E402
from gotennet_amd.graph import distance  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmd17_aspirin")
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--F", type=int, default=256)
ap.add_argument("--L", type=int, default=6)
ap.add_argument("--lmax", type=int, default=2)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()

torch.manual_seed(0)
net = gotennet_amd.GotenNet(n_atom_basis=a.F, n_interactions=a.L, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                            num_heads=8, scale_edge=False, lmax=a.lmax, sep_dir=True, sep_tensor=True).cuda().eval()
net.assume_sorted_edges = True
pos, batch, z = synthetic.make_batch(a.workload, a.batch)
pos, batch, z = pos.cuda(), batch.cuda(), z.cuda()
ei, w, vec = distance(pos, batch, 5.0, 32)
print(f"N={pos.shape[0]} E={ei.shape[1]} F={a.F} L={a.L} lmax={a.lmax}")

events = []
orig_call = _lib.call


def timed_call(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig_call(name, *args)
    e1.record()
    tag = name
    if name == "gn_gemm":
        tag = f"gn_gemm[{args[6]}x{args[7]}x{args[8]}]"
    events.append((tag, e0, e1))


for _ in range(2):
    net(z, ei, w, vec)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.iters):
    net(z, ei, w, vec)
t1.record()
torch.cuda.synchronize()
print(f"forward: {t0.elapsed_time(t1) / a.iters:.3f} ms  ({(a.batch or synthetic.WORKLOADS[a.workload][2]) / (t0.elapsed_time(t1) / a.iters) * 1e3:.0f} mol/s fwd-only)")

engine.call = timed_call
import gotennet_amd.engine as E_  # noqa
E_.call = timed_call
for _ in range(a.iters):
    net(z, ei, w, vec)
torch.cuda.synchronize()
tot = defaultdict(float); cnt = defaultdict(int)
for tag, e0, e1 in events:
    tot[tag] += e0.elapsed_time(e1); cnt[tag] += 1
s = sum(tot.values())
print(f"sum of kernel events per forward: {s / a.iters:.3f} ms")
for tag in sorted(tot, key=tot.get, reverse=True):
    print(f"  {tag:40s} {tot[tag] / a.iters:8.3f} ms/fwd  {cnt[tag] // a.iters:3d} calls  {1e3 * tot[tag] / cnt[tag]:8.1f} us/call  {100 * tot[tag] / s:5.1f}%")
