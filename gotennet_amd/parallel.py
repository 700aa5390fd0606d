"""Molecule-sharded data parallelism: one process per GPU, no data-path collective except ONE
all-reduce of the zero-padded per-molecule energy vector (RCCL over xGMI on MI355X; gloo in the
CPU tests).  Molecules are independent units (reference layers.py:1589: radius_graph(batch=batch))."""
from __future__ import annotations

from collections import deque
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


class OrderedReducer:
    """The per-step energy all-reduce of a run with several batches in flight (pipeline.InFlight), issued from ONE
    communication stream in host SUBMISSION order.

    Each lane of ``InFlight`` runs on its own HIP stream; a collective enqueued "on the lane's stream" would tie the order
    in which a rank's collectives reach the communicator to how its lanes happen to drift.  Here lane k only records
    where its energies become ready (``comm.wait_stream(lane stream)``); shard placement and the all-reduce run on the
    communication stream, call after call, so every rank issues collective #s for submission #s -- the sequence all ranks
    share by construction (round-robin lanes, same number of steps).  ``slots`` result vectors rotate: result #s lives in
    ``bufs[s % slots]`` and is overwritten by submission #s + slots; read it after ``wait()``.

    Without a device stream (CPU tensors: the gloo tests) the same bookkeeping runs synchronously."""

    def __init__(self, n_mol_total: int, first: int, device, group=None, slots: int = 3, dtype=torch.float32):
        self.first, self.n_total, self.group = int(first), int(n_mol_total), group
        self.device = torch.device(device)
        self.bufs = [torch.zeros(n_mol_total, dtype=dtype, device=self.device) for _ in range(max(1, int(slots)))]
        self.comm = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.submitted = 0
        self.order = deque(maxlen=256)               # (submission number, slot) in issue order: what the tests look at

    def submit(self, e_local: torch.Tensor) -> torch.Tensor:
        """Called where ``e_local`` was produced (inside the lane's stream context).  Returns the slot's result vector
        (valid after ``wait()``)."""
        s = self.submitted
        self.submitted += 1
        out = self.bufs[s % len(self.bufs)]
        self.order.append((s, s % len(self.bufs)))
        if self.comm is None:
            reduce_energies(e_local, self.first, self.n_total, group=self.group, out=out)
            return out
        producer = torch.cuda.current_stream(self.device)
        self.comm.wait_stream(producer)              # the energies of this step are ready
        e_local.record_stream(self.comm)
        with torch.cuda.stream(self.comm):           # collective #s follows collective #s-1 on this one stream
            reduce_energies(e_local, self.first, self.n_total, group=self.group, out=out)
        return out

    def wait(self):
        """The current stream waits for every collective submitted so far."""
        if self.comm is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm)
