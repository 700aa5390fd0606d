#!/usr/bin/env python
"""gn_attn_softmax on t_attn rows embedded in the [E, (1+M)F] edge projection (stride 6 KiB at C2) vs a contiguous [E, F]
tensor: is the 1-KiB-of-every-6 access pattern what holds the kernel at ~2.4 TB/s?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import engine, synthetic  # noqa: E402
from gotennet_amd._lib import call, ptr  # noqa: E402
from gotennet_amd.graph import distance  # noqa: E402

dev = "cuda"
pos, batch, z = synthetic.make_batch("rmd17_aspirin", 128, seed=0)
ei, ed, ev = distance(pos.to(dev), batch.to(dev), 5.0, 32)
N, E, F, H, M = pos.shape[0], ei.shape[1], 256, 8, 5
src, dst = ei[0].int().contiguous(), ei[1].int().contiguous()
rowptr = torch.zeros(N + 1, dtype=torch.int32, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=N), 0)
nact = torch.randn(N, 4 * F, device=dev)
eproj = torch.randn(E, (1 + M) * F, device=dev)
ta = eproj[:, :F].contiguous()
a = torch.empty(E, H, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(t, ld):
    call("gn_attn_softmax", ptr(nact), nact.data_ptr() + 4 * F, 4 * F, ptr(t), ld, ptr(rowptr), ptr(src), None, N, F, H, ptr(a), 0, st)


for name, t, ld in (("embedded (ld 1536)", eproj, (1 + M) * F), ("contiguous (ld 256)", ta, F)):
    for _ in range(5):
        run(t, ld)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run(t, ld)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:22s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
