"""Molecule-sharded data parallelism: one process per GPU, no data-path collective except ONE
all-reduce of the zero-padded per-molecule energy vector (RCCL over xGMI on MI355X; gloo in the
CPU tests).  Molecules are independent units (reference layers.py:1589: radius_graph(batch=batch))."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(rank: int, world: int, n_mol_total: int) -> Tuple[int, int]:
    """Contiguous block of the batch index owned by ``rank`` (first, count); remainder goes to the low ranks."""
    base, rem = divmod(n_mol_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def reduce_energies(e_local: torch.Tensor, first: int, n_mol_total: int, group=None,
                    out: torch.Tensor | None = None) -> torch.Tensor:
    """All ranks get the full [n_mol_total] energy vector: each rank writes its shard into a zeroed
    vector, then one SUM all-reduce."""
    import torch.distributed as dist
    if out is None:
        out = torch.zeros(n_mol_total, dtype=e_local.dtype, device=e_local.device)
    else:
        out.zero_()
    out[first:first + e_local.numel()] = e_local.reshape(-1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out
