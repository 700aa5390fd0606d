#!/usr/bin/env python
"""Does overlapping two half-batches help now that the projections sit at the power cap?  Static topology, hipGraph replays
(no host in the loop): one CapturedStep of B molecules vs two of B/2 replayed on two streams.   python tools/two_stream_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gotennet_amd  # noqa: E402
from gotennet_amd import synthetic  # noqa: E402
from gotennet_amd.graph import distance  # noqa: E402
from gotennet_amd.outputs import Atomwise  # noqa: E402
from gotennet_amd.pipeline import CapturedStep, EnergyForces  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                            num_heads=8, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()


def make(B, first):
    pos, batch, z = (v.to(dev) for v in synthetic.make_batch("rmd17_aspirin", B, seed=0, first_molecule=first))
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    return CapturedStep(EnergyForces(rep, head, check_edges=False), z, ei, batch, B), pos


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


full, pos_full = make(128, 0)
t_full = bench(lambda: full(pos_full))
(a, pa), (b, pb) = make(64, 0), make(64, 64)
t_half = bench(lambda: a(pa))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        a.pos.copy_(pa); a.graph.replay()
    with torch.cuda.stream(s2):
        b.pos.copy_(pb); b.graph.replay()


t_two = bench(both)


def serial():
    a(pa); b(pb)


t_ser = bench(serial)
(c, pc), (d, pd) = make(128, 0), make(128, 128)


def both128():
    with torch.cuda.stream(s1):
        c.pos.copy_(pc); c.graph.replay()
    with torch.cuda.stream(s2):
        d.pos.copy_(pd); d.graph.replay()


t_two128 = bench(both128) / 2
print(f"[two-stream lmax={lmax}] two graphs of 128 on two streams: {t_two128:.3f} ms per batch of 128")
print(f"[two-stream lmax={lmax}] one graph of 128: {t_full:.3f} ms | one graph of 64: {t_half:.3f} ms | two of 64 back to back: "
      f"{t_ser:.3f} ms | two of 64 on two streams: {t_two:.3f} ms")
