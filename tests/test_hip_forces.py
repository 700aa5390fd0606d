"""GPU: energy + forces (hand-written backward) against the reference goldens and the oracle's autograd."""
import types

import pytest
import torch

from tests.golden_util import case_names, load_case, rel_err
from tests.test_hip_parity import _net_from_case, _synthetic

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]

TOL = 1e-4          # north_star: forces within 1e-4 relative (max-norm)
# (aggr = "max" has a forward kernel only: its own test below holds the refusal)
FORCE_CASES = [n for n in case_names() if "shuffled" not in n and "aggr_max" not in n]


def _head_from_case(cfg, head_sd):
    from gotennet_amd.outputs import Atomwise
    head = Atomwise(n_in=cfg["n_atom_basis"], n_hidden=cfg.get("head_hidden", 16), property="property", derivative="forces", activation="silu")
    head.load_state_dict(head_sd, strict=True)
    return head.cuda().eval()


@pytest.mark.parametrize("name", FORCE_CASES)
def test_fused_pipeline_matches_golden(name):
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case(name)
    net, head = _net_from_case(cfg, sd), _head_from_case(cfg, head_sd)
    ef = EnergyForces(net, head)
    e, f = ef(t["z"].cuda(), t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda(),
              t["batch"].cuda(), cfg["n_mol"])
    torch.cuda.synchronize()
    assert rel_err(e.cpu(), t["energy"]) < TOL
    assert rel_err(f.cpu(), t["forces"]) < TOL
    # against fp64 truth: not meaningfully worse than the fp32 reference itself
    assert rel_err(f.cpu(), t["forces_f64"]) < max(10 * rel_err(t["forces"], t["forces_f64"]), 2e-5)
    e2, f2 = ef(t["z"].cuda(), t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda(),
                t["batch"].cuda(), cfg["n_mol"])
    assert torch.equal(f, f2) and torch.equal(e, e2)             # bit-reproducible


@pytest.mark.parametrize("name", ["l2_sep_f32", "l1_nosep_scale_f32"])
def test_reference_style_autograd_call(name):
    """GotenNetWrapper + Atomwise(derivative) used the way GotenModel uses them (autograd.grad wrt pos)."""
    import gotennet_amd
    cfg, sd, head_sd, t = load_case(name)
    net = gotennet_amd.GotenNetWrapper(
        n_atom_basis=cfg["n_atom_basis"], n_interactions=cfg["n_interactions"], n_rbf=cfg["n_rbf"],
        cutoff_fn=gotennet_amd.CosineCutoff(cfg["cutoff"]), max_z=cfg["max_z"], num_heads=cfg["num_heads"],
        scale_edge=cfg["scale_edge"], lmax=cfg["lmax"], sep_dir=cfg["sep_dir"], sep_tensor=cfg["sep_tensor"])
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    head = _head_from_case(cfg, head_sd)
    pos = t["pos"].cuda().requires_grad_(True)
    inp = types.SimpleNamespace(z=t["z"].cuda(), pos=pos, batch=t["batch"].cuda())
    inp.representation, inp.vector_representation = net(inp)
    out = head(inp)
    assert rel_err(out["property"].detach().cpu(), t["energy"]) < TOL
    assert rel_err(out["forces"].detach().cpu(), t["forces"]) < TOL


@pytest.mark.parametrize("name", ["l2_sep_f32", "l2_sep_shuffled_noloop"])
def test_edge_level_autograd_boundary(name):
    """GotenNet.forward with edge_vec/edge_diff requiring grad: gradients of a random
    functional of (h, X) w.r.t. both edge inputs match the oracle's autograd (also for an
    unsorted edge list)."""
    from oracle import gotennet_oracle as orc
    cfg, sd, _, t = load_case(name)
    net = _net_from_case(cfg, sd)
    torch.manual_seed(0)
    wh, wX = torch.randn_like(t["h"]), torch.randn_like(t["X"])
    ev = t["edge_vec"].clone().requires_grad_(True)
    ed = t["edge_diff"].clone().requires_grad_(True)
    h, X = orc.gotennet_forward(sd, cfg, t["z"], t["edge_index"], ed, ev)
    gv_ref, gd_ref = torch.autograd.grad((h * wh).sum() + (X * wX).sum(), [ev, ed])
    evc = t["edge_vec"].cuda().requires_grad_(True)
    edc = t["edge_diff"].cuda().requires_grad_(True)
    hc, Xc = net(t["z"].cuda(), t["edge_index"].cuda(), edc, evc)
    gv, gd = torch.autograd.grad((hc * wh.cuda()).sum() + (Xc * wX.cuda()).sum(), [evc, edc])
    mask = t["edge_index"][0] != t["edge_index"][1]
    assert rel_err(gv.cpu()[mask], gv_ref[mask]) < TOL
    assert rel_err(gd.cpu()[mask], gd_ref[mask]) < TOL


@pytest.mark.parametrize("F,L,lmax", [(128, 2, 2), (256, 2, 2), (64, 2, 4), (64, 3, 3), (256, 2, 3), (256, 2, 4),
                                      (512, 2, 2), (512, 2, 3), (1024, 2, 1)])      # (> 256: a slot spans 2 / 4 waves)
def test_forces_match_oracle_wide(F, L, lmax):
    import gotennet_amd
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    from oracle import gotennet_oracle as orc
    torch.manual_seed(F + lmax)
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=L, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True)
    head = Atomwise(n_in=F, n_hidden=64, derivative="forces", activation="silu")
    with torch.no_grad():
        for m in (net, head):
            for n, p in m.named_parameters():
                if p.dim() == 1:
                    p.uniform_(-0.05, 0.05) if "norm.weight" not in n else p.uniform_(0.9, 1.1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    cfg = orc.default_config(n_atom_basis=F, n_interactions=L, n_rbf=32, num_heads=8, scale_edge=False, lmax=lmax,
                             sep_dir=True, sep_tensor=True)
    pos, batch, z = _synthetic(3, 14, 4.0, seed=F)
    e_ref, f_ref, _ = orc.energy_and_forces({k: v.double() for k, v in sd.items()}, cfg,
                                            {k: v.double() for k, v in hsd.items()}, z, pos.double(), batch, 3)
    net, head = net.cuda().eval(), head.cuda().eval()
    ei, w, vec = distance(pos.cuda(), batch.cuda(), 5.0, 32)
    e, f = EnergyForces(net, head)(z.cuda(), ei, w, vec, batch.cuda(), 3)
    assert rel_err(e.cpu(), e_ref) < TOL
    assert rel_err(f.cpu(), f_ref) < TOL
    # conservation: forces of an isolated molecule sum to zero (translation invariance)
    assert float(f.cpu().reshape(3, 14, 3).sum(1).abs().max()) < 1e-3 * float(f.abs().max())


def _random_flag_cases(n=16, lmaxes=(1, 2, 2, 3, 4), rng_seed=1234, first_seed=100):
    """Seeded random flag combinations across every constructor switch the HIP path implements."""
    import random
    rng = random.Random(rng_seed)
    cases = []
    for i in range(n):
        lmax = rng.choice(list(lmaxes))
        parts = []
        if rng.random() < 0.4:
            parts.append(rng.choice(["gated", "gatedt", "act"]))
        if rng.random() < 0.3:
            parts.append("norej")
        if rng.random() < 0.35:
            parts.append(rng.choice(["mlp", "mlpa"]))
        lin = rng.random() < 0.35
        if lin:
            parts.append(rng.choice(["linw", "linwa"]))
            if rng.random() < 0.5:
                parts.append(rng.choice(["ln", "postln"]))
        eu = "_".join(parts) if parts else rng.choice([True, True, False])
        cases.append(dict(
            n_atom_basis=rng.choice([32, 64]), n_interactions=rng.choice([1, 2, 3]), n_rbf=rng.choice([8, 16]), lmax=lmax,
            num_heads=rng.choice([4, 8]), scale_edge=rng.random() < 0.5, sep_dir=rng.random() < 0.6,
            sep_tensor=rng.random() < 0.6, sep_htr=rng.random() < 0.6,
            radial_basis=rng.choice(["expnorm", "expnorm", "BesselBasis", "GaussianRBF"]), edge_updates=eu,
            layernorm=rng.choice(["", "", "layer"]), steerable_norm=rng.choice(["", "", "tensor"]),
            edge_ln=rng.choice(["", "layer"]), evec_dim=(16 if lin and rng.random() < 0.4 else None),
            emlp_dim=rng.choice([None, 48]), seed=first_seed + i,
            activation=rng.choice(["silu", "silu", "softplus", "tanh", "elu", "selu", "mish", "gelu", "sigmoid"])))
    return cases


# the second set crosses the same switches at degrees 5..8 (the degree-sliced kernels of gn_highl.hip)
@pytest.mark.parametrize("hp", _random_flag_cases() + _random_flag_cases(10, (5, 6, 7, 8), 4321, 300),
                         ids=lambda hp: f"s{hp['seed']}_l{hp['lmax']}")
def test_random_flag_combinations_match_oracle(hp):
    """Forward (h, X) and energy/forces against the oracle for random combinations of the constructor flags (the
    golden fixtures pin each flag against the reference; this crosses them)."""
    import gotennet_amd
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    from oracle import gotennet_oracle as orc
    hp = dict(hp)
    seed = hp.pop("seed")
    torch.manual_seed(seed)
    F = hp["n_atom_basis"]
    net = gotennet_amd.GotenNet(cutoff_fn=gotennet_amd.CosineCutoff(5.0), max_z=10, **hp)
    head = Atomwise(n_in=F, n_hidden=32, derivative="forces")             # reference default head activation: shifted softplus
    with torch.no_grad():
        for m in (net, head):
            for n, p in m.named_parameters():
                if p.dim() == 1:
                    p.uniform_(-0.05, 0.05) if not n.endswith("norm.weight") else p.uniform_(0.9, 1.1)
        for n, b in net.named_buffers():
            if n.endswith("tensor_layernorm.weight"):
                b.uniform_(0.9, 1.1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    cfg = orc.default_config(cutoff=5.0, **{k: v for k, v in hp.items() if k not in ("evec_dim", "emlp_dim", "edge_ln")})
    pos, batch, z = _synthetic(2, 9, 3.2, seed=seed)
    z = z.clamp(max=9)
    ei, w, vec = orc.distance(pos, batch, 5.0)
    h_ref, X_ref = orc.gotennet_forward({k: v.double() for k, v in sd.items()}, cfg, z, ei, w.double(), vec.double())
    e_ref, f_ref, _ = orc.energy_and_forces({k: v.double() for k, v in sd.items()}, cfg,
                                            {k: v.double() for k, v in hsd.items()}, z, pos.double(), batch, 2,
                                            activation="softplus")
    net, head = net.cuda().eval(), head.cuda().eval()
    eic, wc, vecc = distance(pos.cuda(), batch.cuda(), 5.0, 32)
    assert torch.equal(eic.cpu(), ei)
    h, X = net(z.cuda(), eic, wc, vecc)
    assert rel_err(h.cpu(), h_ref) < TOL and rel_err(X.cpu(), X_ref) < TOL
    e, f = EnergyForces(net, head)(z.cuda(), eic, wc, vecc, batch.cuda(), 2)
    assert rel_err(e.cpu(), e_ref) < TOL
    assert rel_err(f.cpu(), f_ref) < TOL


def test_full_size_forward_matches_reference():
    """BASELINE configs[1] at FULL size against the reference itself (tests/golden/c2_full_forward_seeded.npz: the
    reference's CPU forward on the bench inputs with the seeded C2 model): bit-exact edge list, sampled rows and column
    sums of (h, X), every molecule's energy."""
    import json
    import numpy as np
    from gotennet_amd import synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces
    from tests.golden_util import GOLDEN_DIR, seeded_modules
    import os
    zf = np.load(os.path.join(GOLDEN_DIR, "c2_full_forward_seeded.npz"))
    cfg = json.loads(bytes(zf["cfg"]).decode())
    net, head = seeded_modules(cfg)
    net, head = net.cuda().eval(), head.cuda().eval()
    pos, batch, z = synthetic.make_batch(cfg["workload"], cfg["n_mol"], seed=cfg["batch_seed"])
    pos, batch, z = pos.cuda(), batch.cuda(), z.cuda()
    ei, ed, ev = distance(pos, batch, cfg["cutoff"], 32)
    assert ei.shape[1] == int(zf["n_edges"])
    chk = [int(ei[0].sum()), int(ei[1].sum()), int((ei[0] * 31 + ei[1]).sum() % (2 ** 61))]
    assert chk == [int(v) for v in zf["edge_index_checksum"]]                # integer work: bit-exact
    h, X = net(z, ei, ed, ev)
    t = lambda k: torch.from_numpy(zf[k])
    assert rel_err(h[t("rows_h").cuda()].cpu(), t("h_rows")) < TOL
    assert rel_err(X[t("rows_X").cuda()].cpu(), t("X_rows")) < TOL
    # column sums over 2688 atoms: errors relative to the absolute mass of the column sums' terms
    assert float((h.double().sum(0).cpu() - t("h_colsum")).abs().max()) < TOL * float(zf["h_abs_sum"]) / h.shape[1]
    assert float((X.double().sum(0).cpu() - t("X_colsum")).abs().max()) < TOL * float(zf["X_abs_sum"]) / X[0].numel()
    e, f = EnergyForces(net, head)(z, ei, ed, ev, batch, cfg["n_mol"])
    assert rel_err(e.cpu(), t("energy")) < TOL
    # forces of the first 32 molecules by the reference's autograd (a quarter batch fits the build container)
    nf = int(zf["n_force_molecules"]) * (pos.shape[0] // cfg["n_mol"])
    assert rel_err(f[:nf].cpu(), t("forces_part")) < TOL


def test_full_size_properties():
    """BASELINE configs[1] at full size (128 aspirin-like molecules, F=256, L=6, lmax=2, the bench model): properties that
    need no reference -- bit-reproducibility, zero net force per molecule, rotation/translation behaviour of E and F,
    per-molecule results independent of what else is in the batch."""
    import gotennet_amd
    from gotennet_amd import synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    dev = "cuda"
    torch.manual_seed(0)
    rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=False, lmax=2, sep_dir=True, sep_tensor=True).to(dev).eval()
    head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
    run = EnergyForces(rep, head)
    B, na = 128, 21
    pos, batch, z = synthetic.make_batch("rmd17_aspirin", B, seed=0)
    pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)

    def ef(p, zz=z, bb=batch, nb=B):
        ei, ed, ev = distance(p, bb, 5.0, 32)
        e, f = run(zz, ei, ed, ev, bb, nb)
        return e.clone(), f.clone()

    e0, f0 = ef(pos)
    assert torch.isfinite(e0).all() and torch.isfinite(f0).all()
    e1, f1 = ef(pos)
    assert torch.equal(e0, e1) and torch.equal(f0, f1)                       # no atomics anywhere: bit-reproducible
    fmax = float(f0.abs().max())
    assert float(f0.reshape(B, na, 3).sum(1).abs().max()) < 2e-4 * fmax       # translation invariance: zero net force
    # rotation + translation of every molecule: E invariant, F co-rotates
    g = torch.Generator().manual_seed(5)
    Q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    Q = (Q * torch.sign(torch.linalg.det(Q))).float().to(dev)
    e2, f2 = ef(pos @ Q.T + torch.tensor([1.5, -2.0, 0.25], device=dev))
    assert rel_err(e2.cpu(), e0.cpu()) < 1e-5
    assert rel_err(f2.cpu(), (f0 @ Q.T).cpu()) < TOL
    # a molecule's energy / forces do not depend on its batch mates: reversed molecule order, and a 5-molecule sub-batch
    perm = torch.arange(B - 1, -1, -1, device=dev)
    idx = (perm[:, None] * na + torch.arange(na, device=dev)[None]).reshape(-1)
    e3, f3 = ef(pos[idx], z[idx], batch)
    assert rel_err(e3.cpu(), e0[perm].cpu()) < 1e-6 and rel_err(f3.cpu(), f0[idx].cpu()) < 1e-5
    sub = slice(7 * na, 12 * na)
    e4, f4 = ef(pos[sub], z[sub], batch[: 5 * na], 5)
    assert rel_err(e4.cpu(), e0[7:12].cpu()) < 1e-6 and rel_err(f4.cpu(), f0[sub].cpu()) < 1e-5


@pytest.mark.parametrize("n_mol,lmax", [(1, 2), (5, 2), (2, 4)])
def test_captured_step_replays_bit_identical(n_mol, lmax):
    """pipeline.CapturedStep (static topology, one hipGraph replay per step) against the eager fused path on the same
    edge list, for several position sets: bit-identical energies and forces."""
    import gotennet_amd
    from gotennet_amd import synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import CapturedStep, EnergyForces
    dev = "cuda"
    torch.manual_seed(3)
    rep = gotennet_amd.GotenNet(n_atom_basis=64, n_interactions=3, n_rbf=16, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=True, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
    head = Atomwise(n_in=64, n_hidden=32, derivative="forces", activation="silu").to(dev).eval()
    ef = EnergyForces(rep, head)
    pos, batch, z = synthetic.make_batch("rmd17_aspirin", n_mol, seed=4)
    pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    step = CapturedStep(ef, z, ei, batch, n_mol)
    g = torch.Generator().manual_seed(9)
    for k in range(4):
        p = pos + 0.05 * k * torch.randn(pos.shape, generator=g).to(dev)      # small moves: the edge list stays valid
        e_g, f_g = step(p)
        e_g, f_g = e_g.clone(), f_g.clone()
        src, dst = ei[0], ei[1]
        vec = p[src] - p[dst]
        diff = torch.where(src == dst, torch.zeros_like(vec[:, 0]), torch.sqrt(vec[:, 0] * vec[:, 0] + vec[:, 1] * vec[:, 1] + vec[:, 2] * vec[:, 2]))
        e_e, f_e = ef(z, ei, diff, vec, batch, n_mol)
        assert torch.equal(e_g, e_e) and torch.equal(f_g, f_e), k


def test_forces_asymmetric_graph_neighbor_cap():
    """A dense molecule with max_num_neighbors far below the neighbour count: the capped radius graph is NOT
    symmetric (j->i present, i->j absent), so the by-source (CSC) backward pass sees different rows than the
    by-target pass.  Same first-k rule as the oracle; forces vs the oracle's fp64 autograd."""
    import gotennet_amd
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    from oracle import gotennet_oracle as orc
    torch.manual_seed(11)
    F, L, lmax, cap = 64, 2, 2, 8
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=L, n_rbf=16, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=True, lmax=lmax, sep_dir=True, sep_tensor=True)
    head = Atomwise(n_in=F, n_hidden=32, derivative="forces", activation="silu")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    cfg = orc.default_config(n_atom_basis=F, n_interactions=L, n_rbf=16, num_heads=8, scale_edge=True, lmax=lmax,
                             sep_dir=True, sep_tensor=True)
    g = torch.Generator().manual_seed(5)
    pos = torch.rand((60, 3), generator=g) * 6.0
    batch = torch.zeros(60, dtype=torch.long)
    z = torch.randint(1, 9, (60,), generator=g)
    ei_ref = orc.radius_graph(pos, batch, 5.0, cap, loop=True)
    src, dst = ei_ref
    pairs = set(zip(src.tolist(), dst.tolist()))
    assert any((d, s_) not in pairs for s_, d in pairs)          # really asymmetric
    e_ref, f_ref, _ = orc.energy_and_forces({k: v.double() for k, v in sd.items()}, cfg,
                                            {k: v.double() for k, v in hsd.items()}, z, pos.double(), batch, 1,
                                            max_num_neighbors=cap)
    net, head = net.cuda().eval(), head.cuda().eval()
    ei, w, vec = distance(pos.cuda(), batch.cuda(), 5.0, cap)
    assert torch.equal(ei.cpu(), ei_ref)
    e, f = EnergyForces(net, head)(z.cuda(), ei, w, vec, batch.cuda(), 1)
    assert rel_err(e.cpu(), e_ref) < TOL
    assert rel_err(f.cpu(), f_ref) < TOL


def test_forces_high_degree_cluster():
    """Every atom of a 120-atom cluster inside the cutoff of every other one: 121 incoming and 121 outgoing edges per atom
    (more than a wave of lanes in the position scatter, more than one pass of the per-target slots everywhere, 968 head
    scores per target in the softmax backward).  Energy and forces vs the oracle's fp64 autograd."""
    import gotennet_amd
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    from oracle import gotennet_oracle as orc
    torch.manual_seed(21)
    F, L, lmax, cap, n = 64, 2, 2, 128, 120
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=L, n_rbf=16, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=True, lmax=lmax, sep_dir=True, sep_tensor=True)
    head = Atomwise(n_in=F, n_hidden=32, derivative="forces", activation="silu")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    cfg = orc.default_config(n_atom_basis=F, n_interactions=L, n_rbf=16, num_heads=8, scale_edge=True, lmax=lmax,
                             sep_dir=True, sep_tensor=True)
    g = torch.Generator().manual_seed(6)
    pos = torch.rand((n, 3), generator=g) * 2.5                  # the cube's diagonal is 4.33 < 5
    batch = torch.zeros(n, dtype=torch.long)
    z = torch.randint(1, 9, (n,), generator=g)
    e_ref, f_ref, _ = orc.energy_and_forces({k: v.double() for k, v in sd.items()}, cfg,
                                            {k: v.double() for k, v in hsd.items()}, z, pos.double(), batch, 1,
                                            max_num_neighbors=cap)
    net, head = net.cuda().eval(), head.cuda().eval()
    ei, w, vec = distance(pos.cuda(), batch.cuda(), 5.0, cap)
    assert ei.shape[1] == n * n                                   # complete graph incl. self-loops
    e, f = EnergyForces(net, head)(z.cuda(), ei, w, vec, batch.cuda(), 1)
    assert rel_err(e.cpu(), e_ref) < TOL
    assert rel_err(f.cpu(), f_ref) < TOL
    assert float(f.sum(0).abs().max()) < 1e-3 * float(f.abs().max())


def test_aggr_max_forces_match_reference():
    """aggr='max' (gotennet.py:84,638-639): the element-wise maximum over the incoming messages, and its input-gradient -- the
    upstream gradient of every output element routed to the arg-max edge (hl_max_route_kernel) -- against the reference's
    energies and forces (fixture opt_aggr_max_l3) and the oracle on a second system; bit-reproducible."""
    from oracle import gotennet_oracle as orc
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case("opt_aggr_max_l3")
    net, head = _net_from_case(cfg, sd), _head_from_case(cfg, head_sd)
    assert net.config().aggr == 2
    args = [t[k].cuda() for k in ("z", "edge_index", "edge_diff", "edge_vec", "batch")] + [cfg["n_mol"]]
    e, _ = EnergyForces(net, head)(*args, forces=False)
    assert rel_err(e.cpu(), t["energy"]) < 1e-4
    e, f = EnergyForces(net, head)(*args)
    assert rel_err(e.cpu(), t["energy"]) < 1e-4 and rel_err(f.cpu(), t["forces"]) < TOL
    e2, f2 = EnergyForces(net, head)(*args)
    assert torch.equal(e, e2) and torch.equal(f, f2)
    # another geometry of the same model, against the oracle's autograd through scatter_reduce(amax)
    g = torch.Generator().manual_seed(4)
    pos = t["pos"] + 0.05 * torch.randn(t["pos"].shape, generator=g)
    e_ref, f_ref, _ = orc.energy_and_forces(sd, cfg, head_sd, t["z"], pos, t["batch"], cfg["n_mol"])
    from gotennet_amd.graph import distance
    ei, w, vec = distance(pos.cuda(), t["batch"].cuda(), cfg["cutoff"], 32)
    e, f = EnergyForces(net, head)(t["z"].cuda(), ei, w, vec, t["batch"].cuda(), cfg["n_mol"])
    assert rel_err(e.cpu(), e_ref) < 1e-4 and rel_err(f.cpu(), f_ref) < TOL
