#!/usr/bin/env python
"""Segment-resident K7 (gn_htr_edge_seg: node rows of a molecule in LDS) against the per-target kernel (gn_htr_edge):
bit-equality of the weights and time per launch, C2 shapes, lmax from argv."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import synthetic  # noqa: E402
from gotennet_amd._lib import call, ptr  # noqa: E402
from gotennet_amd.graph import distance  # noqa: E402

dev = "cuda"
lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D = (lmax + 1) ** 2 - 1
pos, batch, z = synthetic.make_batch("rmd17_aspirin", 128, seed=0)
ei, ed, ev = distance(pos.to(dev), batch.to(dev), 5.0, 32)
N, E, F = pos.shape[0], ei.shape[1], 256
order = torch.argsort(ei[1], stable=True)
src, dst = ei[0][order].int().contiguous(), ei[1][order].int().contiguous()
rowptr = torch.zeros(N + 1, dtype=torch.int32, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=N), 0)
b = batch.to(dev)
cnt = torch.bincount(b)
mp = torch.zeros(cnt.numel() + 1, dtype=torch.long, device=dev); mp[1:] = torch.cumsum(cnt, 0)
seg_hi = mp[b + 1].int().contiguous()
seg_first = mp[:-1].int().contiguous()
nseg = torch.tensor([cnt.numel()], dtype=torch.int32, device=dev)
cap = int(cnt.max())
EQ, EK = torch.randn(N, D, F, device=dev), torch.randn(N, D, F, device=dev)
rl = torch.randn(E, D, device=dev)
w0, w1 = torch.empty(E, F, device=dev), torch.zeros(E, F, device=dev)
st = torch.cuda.current_stream().cuda_stream


def old():
    call("gn_htr_edge", ptr(EQ), ptr(EK), ptr(rl), ptr(rowptr), ptr(src), N, F, lmax, 0, None, ptr(w0), st)


def new(c=cap):
    call("gn_htr_edge_seg", ptr(EQ), ptr(EK), ptr(rl), ptr(rowptr), ptr(src), ptr(dst), ptr(seg_first), ptr(seg_hi), ptr(nseg),
         N, F, lmax, c, ptr(w1), st)


old(); new(); torch.cuda.synchronize()
print("N", N, "E", E, "lmax", lmax, "cap", cap, "bit-equal:", bool(torch.equal(w0, w1)), "maxdiff", float((w0 - w1).abs().max()))
for name, fn in (("per-target", old), ("segment", new)):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:12s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
