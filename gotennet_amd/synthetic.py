"""Synthetic molecule batches of the shapes BASELINE.json names.

Input synthesis only (positions, atomic numbers, batch index) -- no path arithmetic."""
from __future__ import annotations

import torch

# name -> (atoms per molecule, cube side in Angstrom, default batch)
WORKLOADS = {
    "qm9_small": (19, 4.0, 1),            # C1
    "rmd17_aspirin": (21, 4.6, 128),      # C2 / C4 per GPU
    "md22_ac_ala3": (42, 6.0, 64),        # C3
    "md22_nanotube": (370, 14.5, 8),      # C5
}
ELEMENTS = torch.tensor([1, 6, 7, 8])


def make_batch(workload: str, n_mol: int | None = None, seed: int = 0, first_molecule: int = 0):
    """-> pos [N,3] fp32, batch [N] int64, z [N] int64 (CPU tensors).

    Molecule m (global index first_molecule + m) is seeded with ``seed + global index`` so a
    rank's shard is the same set of molecules whatever the world size."""
    n_atoms, side, default_b = WORKLOADS[workload]
    n_mol = default_b if n_mol is None else n_mol
    pos, z = [], []
    for m in range(n_mol):
        g = torch.Generator().manual_seed(seed + first_molecule + m)
        pos.append(torch.rand((n_atoms, 3), generator=g) * side)
        z.append(ELEMENTS[torch.randint(0, 4, (n_atoms,), generator=g)])
    batch = torch.arange(n_mol).repeat_interleave(n_atoms)
    return torch.cat(pos), batch, torch.cat(z)
