"""BASELINE configs[2] (C3: MD22 Ac-Ala3-NHMe-like, batch 64, forces) and configs[4] (C5: MD22 double-walled-nanotube-like,
batch 8, lmax=3, 32-neighbour cap active).

Two layers of evidence:
* reference fixtures ``tests/golden/c3_*.npz`` / ``c5_*.npz`` (tools/make_golden.py: the real reference on molecules of
  the bench workload, radius graph by the reference's Distance.forward, layers.py:1588-1604, so the neighbour-cap rule
  is pinned): CPU -> the oracle against them; GPU -> the HIP path against them (edge list bit-exact);
* GPU, FULL workload size: properties that need no reference (bit-reproducible, zero net force, molecules independent
  of their batch mates, translation; rotation of E / F for C3 only -- the reference is not rotation-invariant for
  lmax >= 3), and molecule 0.. of the full batch equal to the fixture's reference values.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN_DIR, rel_err, seeded_modules

TOL = 1e-4
FIXTURES = ["c3_ac_ala3_2mol_seeded", "c5_nanotube_1mol_seeded"]


def engine_mode():
    from gotennet_amd import engine
    return engine.GEMM_MODE


def _load(name):
    zf = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = json.loads(bytes(zf["cfg"]).decode())
    return cfg, {k: torch.from_numpy(zf[k]) for k in zf.files if k != "cfg"}


def _inputs(cfg, n_mol=None):
    from gotennet_amd import synthetic
    return synthetic.make_batch(cfg["workload"], cfg["n_mol"] if n_mol is None else n_mol, seed=cfg["batch_seed"])


# ------------------------------------------------------------------------------------------ CPU: oracle vs reference
@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_matches_reference_on_workload(name):
    from oracle import gotennet_oracle as orc
    cfg, t = _load(name)
    net, head = seeded_modules(cfg)
    sd, hsd = net.state_dict(), head.state_dict()
    pos, batch, z = _inputs(cfg)
    ei, w, vec = orc.distance(pos, batch, cfg["cutoff"])
    assert torch.equal(ei, t["edge_index"].long())                      # neighbour cap: same first-k rule as Distance
    assert int(torch.bincount(ei[1]).max()) == int(t["max_in_degree"]) == 32
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    if pos.shape[0] <= 200:                               # C3: energy + forces through autograd
        e, f, (h, X, _) = orc.energy_and_forces(sd, cfg, hsd, z, pos, batch, cfg["n_mol"])
        assert rel_err(f, t["forces"]) < 5e-5
    else:                                                                # C5: forward + energy (forces: 15 GB of autograd state)
        with torch.no_grad():
            h, X = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
            e = orc.atomwise_energy(hsd, h, batch, cfg["n_mol"])
    assert rel_err(e.reshape(-1), t["energy"].reshape(-1)) < 2e-5
    assert rel_err(h[t["rows_h"]], t["h_rows"]) < 2e-5
    assert rel_err(X[t["rows_X"]], t["X_rows"]) < 2e-5


# ------------------------------------------------------------------------------------------ GPU: HIP vs reference
def _gpu_modules(cfg):
    net, head = seeded_modules(cfg)
    return net.cuda().eval(), head.cuda().eval()


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
@pytest.mark.parametrize("name", FIXTURES)
def test_hip_matches_reference_on_workload(name):
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces
    cfg, t = _load(name)
    net, head = _gpu_modules(cfg)
    pos, batch, z = (v.cuda() for v in _inputs(cfg))
    ei, ed, ev = distance(pos, batch, cfg["cutoff"], 32)
    assert torch.equal(ei.cpu(), t["edge_index"].long())                 # edge_index bit-exact, cap active
    h, X = net(z, ei, ed, ev)
    assert rel_err(h[t["rows_h"].cuda()].cpu(), t["h_rows"]) < TOL
    assert rel_err(X[t["rows_X"].cuda()].cpu(), t["X_rows"]) < TOL
    e_ref = max(rel_err(t["h_rows"], t["h_rows_f64"]), rel_err(t["X_rows"], t["X_rows_f64"]))
    e_hip = max(rel_err(h[t["rows_h"].cuda()].cpu(), t["h_rows_f64"]), rel_err(X[t["rows_X"].cuda()].cpu(), t["X_rows_f64"]))
    assert e_hip < max(10 * e_ref, 1e-5)                                 # vs fp64 truth: not worse than the fp32 reference
    assert float((h.double().sum(0).cpu() - t["h_colsum"]).abs().max()) < TOL * float(t["h_abs_sum"]) / h.shape[1]
    assert float((X.double().sum(0).cpu() - t["X_colsum"]).abs().max()) < TOL * float(t["X_abs_sum"]) / X[0].numel()
    e, f = EnergyForces(net, head)(z, ei, ed, ev, batch, cfg["n_mol"])
    assert rel_err(e.cpu().reshape(-1), t["energy"].reshape(-1)) < TOL
    assert rel_err(f.cpu(), t["forces"]) < TOL


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
@pytest.mark.parametrize("name,n_mol", [("c3_ac_ala3_2mol_seeded", 64), ("c5_nanotube_1mol_seeded", 8)])
def test_full_size_workload_properties(name, n_mol):
    """The BASELINE batch (C3: 64 x 42 atoms, E ~ 78 k; C5: 8 x 370 atoms, E ~ 89 k, lmax = 3) with the fixture's model."""
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces
    cfg, t = _load(name)
    net, head = _gpu_modules(cfg)
    run = EnergyForces(net, head)
    pos, batch, z = (v.cuda() for v in _inputs(cfg, n_mol))
    na = pos.shape[0] // n_mol
    ei, ed, ev = distance(pos, batch, cfg["cutoff"], 32)
    deg = torch.bincount(ei[1], minlength=pos.shape[0])
    assert int(deg.max()) == 32 and int((deg == 32).sum()) > 0.25 * pos.shape[0]     # the cap really is active
    e0, f0 = (v.clone() for v in run(z, ei, ed, ev, batch, n_mol))
    assert torch.isfinite(e0).all() and torch.isfinite(f0).all()
    e1, f1 = run(z, ei, ed, ev, batch, n_mol)
    assert torch.equal(e0, e1) and torch.equal(f0, f1)                               # bit-reproducible
    fmax = float(f0.abs().max())
    assert float(f0.reshape(n_mol, na, 3).sum(1).abs().max()) < 5e-4 * fmax           # zero net force per molecule
    # the first molecules of the full batch ARE the fixture's molecules: reference energies / forces
    nm = cfg["n_mol"]
    assert rel_err(e0[:nm].cpu().reshape(-1), t["energy"].reshape(-1)) < TOL
    assert rel_err(f0[: nm * na].cpu(), t["forces"]) < TOL
    # translation (graph rebuilt from the moved positions: distances change only by rounding)
    shift = torch.tensor([0.75, -1.25, 2.0], device="cuda")
    ei2, ed2, ev2 = distance(pos + shift, batch, cfg["cutoff"], 32)
    if torch.equal(ei2, ei):
        e2, f2 = run(z, ei2, ed2, ev2, batch, n_mol)
        assert rel_err(e2.cpu(), e0.cpu()) < 1e-5 and rel_err(f2.cpu(), f0.cpu()) < TOL
    # molecules do not depend on their batch mates: reversed molecule order, and a sub-batch
    perm = torch.arange(n_mol - 1, -1, -1, device="cuda")
    idx = (perm[:, None] * na + torch.arange(na, device="cuda")[None]).reshape(-1)
    ei3, ed3, ev3 = distance(pos[idx], batch, cfg["cutoff"], 32)
    e3, f3 = run(z[idx], ei3, ed3, ev3, batch, n_mol)
    # (the fp16 block-exponent arithmetic rounds a row differently when its 8-row block changes: energies move at 1e-6)
    etol = 5e-6 if engine_mode() == "f16x2" else 1e-6
    assert rel_err(e3.cpu(), e0[perm].cpu()) < etol and rel_err(f3.cpu(), f0[idx].cpu()) < 1e-5
    k0, k1 = n_mol // 2, n_mol // 2 + 2
    sub = slice(k0 * na, k1 * na)
    ei4, ed4, ev4 = distance(pos[sub], batch[: 2 * na], cfg["cutoff"], 32)
    e4, f4 = run(z[sub], ei4, ed4, ev4, batch[: 2 * na], 2)
    assert rel_err(e4.cpu(), e0[k0:k1].cpu()) < etol and rel_err(f4.cpu(), f0[sub].cpu()) < 1e-5
    if cfg["lmax"] <= 2:
        # rotation about each molecule's frame on the SAME edge list (the cap's first-k choice is index-based):
        # E invariant, F co-rotates.  Not valid for lmax >= 3 (the reference's l >= 3 harmonics are not normalised).
        g = torch.Generator().manual_seed(5)
        Q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
        Q = (Q * torch.sign(torch.linalg.det(Q))).float().cuda()
        e5, f5 = run(z, ei, ed, ev @ Q.T, batch, n_mol)
        assert rel_err(e5.cpu(), e0.cpu()) < 1e-5
        assert rel_err(f5.cpu(), (f0 @ Q.T).cpu()) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["split", "f16x2"])
def test_large_batch_is_batch_independent(mode):
    """C4's GLOBAL batch (1024 molecules, E = 433 684) on ONE GPU: the first 128 molecules give the 128-molecule batch's
    energies and forces -- bit for bit in the bf16-split arithmetic (no index arithmetic depends on the batch size;
    tools/big_batch_check.py takes the same check to 4096 molecules, where E (1+M) F exceeds 2^31 elements, 143 GiB of
    the 288), and to rounding in the fp16 block-exponent arithmetic (a larger batch switches some products to the
    128-row tile, whose 8-row blocks share exponents differently)."""
    import gotennet_amd
    from gotennet_amd import engine, synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    torch.manual_seed(0)
    net = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=False, lmax=2, sep_dir=True, sep_tensor=True).cuda().eval()
    head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").cuda().eval()
    ef = EnergyForces(net, head)
    old, engine.GEMM_MODE = engine.GEMM_MODE, mode
    try:
        out = {}
        for B in (128, 1024):
            pos, batch, z = synthetic.make_batch("rmd17_aspirin", B, seed=0)
            ei, w, vec = distance(pos.cuda(), batch.cuda(), 5.0, 32)
            out[B] = ef(z.cuda(), ei, w, vec, batch.cuda(), B)
            torch.cuda.synchronize()
    finally:
        engine.GEMM_MODE = old
    e, f = out[1024]
    if mode == "split":
        assert torch.equal(e[:128], out[128][0]) and torch.equal(f[:128 * 21], out[128][1])
    else:
        assert rel_err(e[:128].cpu(), out[128][0].cpu()) < 5e-6 and rel_err(f[:128 * 21].cpu(), out[128][1].cpu()) < 1e-5
    assert bool(torch.isfinite(e).all()) and bool(torch.isfinite(f).all())
    assert float(f.reshape(1024, 21, 3).sum(1).abs().max()) < 1e-3 * float(f.abs().max())
    del out, e, f
    torch.cuda.empty_cache()
