#!/usr/bin/env python
"""A/B of the inference forward (energy only) with gn_message_fused vs the three-kernel sequence, per workload, in one
process: python tools/fused_ab.py [mode]   (prints ms/step for both and the fused kernel's per-launch time)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gotennet_amd  # noqa: E402
from gotennet_amd import _lib, synthetic  # noqa: E402
from gotennet_amd.graph import distance  # noqa: E402
from gotennet_amd.outputs import Atomwise, molecule_ptr  # noqa: E402
from gotennet_amd.pipeline import EnergyForces  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
dev = torch.device("cuda")


class Timer:
    def __init__(self):
        self.events = []

    def want(self, name, args):
        if name in ("gn_message_fused", "gn_message_aggregate", "gn_attn_softmax"):
            return name
        if name.startswith("gn_gemm_group") and args[0][0].M > 40000 and args[0][0].N > 512:
            return "edge_projection"
        return None


for workload, B, lmax in (("rmd17_aspirin", 128, 2), ("rmd17_aspirin", 128, 4), ("md22_ac_ala3", 64, 2), ("md22_nanotube", 8, 3)):
    torch.manual_seed(0)
    rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
    rep.gemm_mode = mode
    head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
    pos, batch, z = (v.to(dev) for v in synthetic.make_batch(workload, B, seed=0))
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    mp = molecule_ptr(batch, B)
    ef = EnergyForces(rep, head, check_edges=False)
    res = {}
    for fused in (True, False):
        rep.fuse_message = fused
        for _ in range(3):
            e, _ = ef(z, ei, ed, ev, batch, B, mol_ptr=mp, forces=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            e, _ = ef(z, ei, ed, ev, batch, B, mol_ptr=mp, forces=False)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 20
        kt = Timer()
        _lib.TIMER = kt
        ef(z, ei, ed, ev, batch, B, mol_ptr=mp, forces=False)
        torch.cuda.synchronize()
        _lib.TIMER = None
        tot = {}
        for tag, e0, e1 in kt.events:
            tot.setdefault(tag, []).append(1e3 * e0.elapsed_time(e1))
        res[fused] = (ms, {k: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for k, v in tot.items()}, e.clone())
    d = float((res[True][2] - res[False][2]).abs().max() / res[False][2].abs().max())
    print(f"{workload} b={B} lmax={lmax} E={ei.shape[1]} mode={mode}: fused {res[True][0]:.3f} ms {res[True][1]} | "
          f"sequence {res[False][0]:.3f} ms {res[False][1]} | energy diff {d:.1e}", flush=True)
