#!/usr/bin/env python
"""Energy+forces throughput with 1 and 3 batches in flight (pipeline.InFlight; eager launches, fresh topology) for the library
GN_LIB_PATH selects -- the A/B loop of round 5 (merged message backward, GEMM grid caps):
   GN_LIB_PATH=gotennet_amd/variants/lib_x.so python tools/lanes_ab.py [workload batch lmax] ; python tools/lanes_ab.py ..."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import gotennet_amd  # noqa: E402
from gotennet_amd.outputs import Atomwise  # noqa: E402

dev = torch.device("cuda")
tag = os.path.basename(os.environ.get("GN_LIB_PATH", "product"))
wl, B, lmax = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("rmd17_aspirin", 128, 2)
torch.manual_seed(0)
rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0), num_heads=8,
                            scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
a = argparse.Namespace(batch=B, workload=wl)
for lanes in (1, 3):
    r = min(bench.in_flight(a, rep, head, dev, lmax, steps=24, lanes=lanes)["ms_per_batch"] for _ in range(3))
    print(f"[{tag} {wl} b={B} lmax={lmax}] lanes {lanes}: {r} ms/batch", flush=True)
