"""Fused energy + force step: representation forward (with tape) -> Atomwise head ->
head gradient -> hand-written backward -> force scatter, all through the C ABI with no
autograd graph and no host synchronisation (hipGraph-capturable).  This is the path
bench.py times; the autograd.Function wrappers in gotennet.py expose the same kernels to
reference-style callers (GotenModel / torch.autograd.grad)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import engine
from .gotennet import GotenNet
from .outputs import Atomwise, molecule_ptr


class EnergyForces:
    def __init__(self, representation: GotenNet, head: Atomwise):
        self.rep, self.head = representation, head

    @torch.no_grad()
    def __call__(self, z: torch.Tensor, edge_index: torch.Tensor, edge_diff: torch.Tensor, edge_vec: torch.Tensor,
                 batch: torch.Tensor, n_mol: int, mol_ptr: Optional[torch.Tensor] = None,
                 forces: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """Edges must be target-sorted (radius-graph order).  -> (energy [n_mol,1], forces [N,3])."""
        rep = self.rep
        cfg, pw = rep.config(), rep.packed_weights()
        N = z.shape[0]
        z32 = z.to(torch.int32)
        g = engine.Graph(cfg, pw, N, edge_index, edge_diff, edge_vec)
        h, X, tape = engine.forward(cfg, pw, z32, g, save=forces)
        if mol_ptr is None:
            mol_ptr = molecule_ptr(batch, n_mol)
        e, y, pre1 = self.head.energy_raw(h, z32, mol_ptr, n_mol)
        if not forces:
            return e, None
        gh = self.head.grad_h_raw(pre1, cfg.F)
        g_vec, g_diff = engine.backward(cfg, pw, z32, g, tape, gh, None)
        return e, engine.pos_gradient(g, g_vec, g_diff, sign=-1.0)
