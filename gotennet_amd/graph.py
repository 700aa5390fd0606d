"""Radius graph + edge vectors on the device (reference Distance.forward,
gotennet/models/components/layers.py:1566-1604, over torch_cluster.radius_graph).

Kernels: csrc/gn_graph.hip.  The only host work is the exclusive scan of the
per-target degrees and the read-back of the edge count needed to size the
outputs (one sync, as in torch_cluster)."""
from __future__ import annotations

import torch

from ._lib import GotenNetHipError, call, ptr


def distance(pos: torch.Tensor, batch: torch.Tensor, cutoff: float, max_num_neighbors: int = 32):
    """-> edge_index int64 [2,E] (row 0 = source j, row 1 = target i; target-major, sources
    ascending, self-loops included), edge_weight [E] (0 on self-loops), edge_vec [E,3] = pos[j]-pos[i]."""
    if not pos.is_cuda:
        raise GotenNetHipError("gotennet_amd.graph.distance runs on a ROCm device only (no CPU fallback)")
    pos = pos.detach().to(torch.float32).contiguous()
    batch = batch.to(torch.int64).contiguous()
    N = pos.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    deg = torch.empty(N, dtype=torch.int32, device=pos.device)
    call("gn_radius_count", ptr(pos), ptr(batch), N, float(cutoff), int(max_num_neighbors), ptr(deg), st)
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=pos.device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    E = int(rowptr[-1].item()) if N else 0
    edge_index = torch.empty((2, E), dtype=torch.int64, device=pos.device)
    edge_vec = torch.empty((E, 3), dtype=torch.float32, device=pos.device)
    edge_diff = torch.empty(E, dtype=torch.float32, device=pos.device)
    call("gn_radius_fill", ptr(pos), ptr(batch), N, float(cutoff), int(max_num_neighbors), ptr(rowptr), E,
         ptr(edge_index), ptr(edge_vec), ptr(edge_diff), st)
    return edge_index, edge_diff, edge_vec
