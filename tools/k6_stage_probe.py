#!/usr/bin/env python
"""GATA message stage (scores + segment softmax + message + aggregate) at the C2 shapes: fused launch vs the
two-launch form, bit-compared and timed (GPU box).  GN_LIB_PATH selects a tuning variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gotennet_amd  # noqa: E402
from gotennet_amd import engine, synthetic  # noqa: E402
from gotennet_amd.graph import distance  # noqa: E402

dev = "cuda"
workload, B = os.environ.get("WL", "rmd17_aspirin"), int(os.environ.get("B", 128))
pos, batch, z = synthetic.make_batch(workload, B, seed=0)
ei, ed, ev = distance(pos.to(dev), batch.to(dev), 5.0, 32)
N, E = pos.shape[0], ei.shape[1]
F, H = 256, 8
for lmax in (2, 3, 4):
    torch.manual_seed(lmax)
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=1, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=H, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
    cfg, pw = net.config(), net.packed_weights()
    g = engine.Graph(cfg, pw, N, ei, ed, ev)
    M, D = cfg.M, cfg.D
    r = lambda *s: torch.randn(*s, device=dev)
    nact, xs, vs, eproj = r(N, 4 * F), r(N, M * F), r(N, M * F), r(E, (1 + M) * F)
    h, X = r(N, F), r(N, D, F)
    outs = {}
    for fuse in (True, False):
        engine.FUSE_ATTENTION = fuse
        attn, h2, X2 = torch.empty(E, H, device=dev), torch.empty(N, F, device=dev), torch.empty(N, D, F, device=dev)
        run = lambda: engine.message_stage(cfg, g, nact, xs, vs, eproj, attn, h, X, h2, X2)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        outs[fuse] = (attn.clone(), h2.clone(), X2.clone())
        nbytes = 4 * N * (2 * F + 2 * M * F + D * F) + E * (4 * (F + M * F + D + 2) + 16) + 4 * N * (F + D * F)
        print(f"lmax={lmax} fused={int(fuse)}: {us:7.1f} us/stage  {nbytes / us / 1e3:7.1f} GB/s = {nbytes / us / 8e6:.3f} of 8 TB/s")
    same = all(torch.equal(a, b) for a, b in zip(outs[True], outs[False]))
    print(f"lmax={lmax}: fused == two-launch bit for bit: {same}")
