"""Boundary behaviour of the drop-in modules: stand-alone GATA / EQFF calls (reference gotennet.py:366-450, 716-748),
caller-supplied edge lists in any order, the Atomwise head with non-trivial mean / stddev / atomref (outputs.py:323-376),
cache invalidation after in-place weight writes, the inference-only warning."""
import os
import warnings

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN_DIR, load_case, rel_err

TOL = 1e-4


def _head_kat():
    k = np.load(os.path.join(GOLDEN_DIR, "kat_head.npz"))
    t = {n: torch.from_numpy(k[n]) for n in k.files if not n.startswith("head/")}
    hsd = {n[5:]: torch.from_numpy(k[n]) for n in k.files if n.startswith("head/")}
    return t, hsd


# --------------------------------------------------------------------------------------------------- CPU
def test_oracle_head_matches_reference_kat():
    """The oracle's Atomwise restatement against the reference's own head with mean / stddev / atomref set."""
    from oracle import gotennet_oracle as orc
    t, hsd = _head_kat()
    y = orc.atomwise_contributions(hsd, t["h"], t["z"])
    e = orc.atomwise_energy(hsd, t["h"], t["batch"], int(t["n_mol"]), z=t["z"])
    assert rel_err(y, t["contrib"]) < 1e-6
    assert rel_err(e, t["energy"]) < 1e-6


def _head_kat3():
    k = np.load(os.path.join(GOLDEN_DIR, "kat_head_l3_mean_ssp.npz"))
    t = {n: torch.from_numpy(k[n]) for n in k.files if not n.startswith("head/")}
    hsd = {n[5:]: torch.from_numpy(k[n]) for n in k.files if n.startswith("head/")}
    return t, hsd


def test_oracle_deep_head_matches_reference_kat():
    """Three-layer pyramid head, the reference's default activation (shifted softplus), aggregation "mean"."""
    from oracle import gotennet_oracle as orc
    t, hsd = _head_kat3()
    y = orc.atomwise_contributions(hsd, t["h"], t["z"], activation="softplus")
    e = orc.atomwise_energy(hsd, t["h"], t["batch"], int(t["n_mol"]), activation="softplus", z=t["z"], aggregation="mean")
    assert rel_err(y, t["contrib"]) < 1e-6
    assert rel_err(e, t["energy"]) < 1e-6


def test_weight_init_names_of_the_reference():
    """Every init name the reference accepts (layers.py:426-452) builds a module (checkpoints carry them as strings)."""
    import gotennet_amd
    torch.manual_seed(0)                                         # (the variance checks below are statistical: fixed draws)
    for name in ("xavier_uniform", "glo_orthogonal", "he_orthogonal", "zeros", ""):
        net = gotennet_amd.GotenNet(n_atom_basis=32, n_interactions=1, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                    lmax=1, weight_init=name)
        assert torch.isfinite(net.gata_list[0].W_q.weight).all()
    with pytest.raises(ValueError):
        gotennet_amd.GotenNet(n_atom_basis=32, n_interactions=1, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                              weight_init="no_such_init")
    w = torch.empty(64, 32)
    from gotennet_amd.layers import glorot_orthogonal_, he_orthogonal_
    glorot_orthogonal_(w)
    assert abs(float(w.var()) - 2.0 / (64 + 32)) < 1e-6          # Glorot variance scale / (fan_in + fan_out)
    he_orthogonal_(w)
    assert abs(float(w.var(dim=1).mean()) - 1.0 / 32) < 1e-3     # standardised rows scaled by 1 / fan_in


def test_packed_cache_invalidation_hooks():
    """Version-counter writes are seen; load_state_dict / reset_parameters / invalidate_packed drop the pack."""
    import gotennet_amd
    net = gotennet_amd.GotenNet(n_atom_basis=32, n_interactions=2, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0), lmax=2)
    pw0 = net.packed_weights()
    assert net.packed_weights() is pw0
    with torch.no_grad():
        net.gata_list[0].W_q.weight.mul_(2.0)                    # bumps _version
    pw1 = net.packed_weights()
    assert pw1 is not pw0 and torch.equal(pw1.layers[0].Wn1[:32], net.gata_list[0].W_q.weight)
    net.gata_list[0].W_q.weight.data.mul_(0.5)                   # .data write: invisible to the version counter ...
    net.invalidate_packed()                                      # ... so the documented hook must be called
    assert torch.equal(net.packed_weights().layers[0].Wn1[:32], net.gata_list[0].W_q.weight)
    pw2 = net.packed_weights()
    net.load_state_dict(net.state_dict())
    assert net.packed_weights() is not pw2
    pw3 = net.packed_weights()
    net.reset_parameters()
    assert net.packed_weights() is not pw3


def test_packed_cache_sees_replaced_parameter_objects():
    """A Parameter OBJECT that is replaced -- torch.func.functional_call, a PARENT module's load_state_dict(assign=True),
    ``layer.weight = nn.Parameter(...)`` -- leaves the old tensors unchanged: the pack is keyed on the identity of every
    slot of the module tree, so the very next call repacks (it used to take up to 32 calls)."""
    import gotennet_amd
    mk = lambda: gotennet_amd.GotenNet(n_atom_basis=32, n_interactions=2, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0), lmax=2)
    net = mk()
    pw0 = net.packed_weights()
    new_w = torch.nn.Parameter(torch.randn(32, 32))
    net.gata_list[0].W_q.weight = new_w                                         # plain attribute replacement
    pw1 = net.packed_weights()
    assert pw1 is not pw0 and torch.equal(pw1.layers[0].Wn1[:32], new_w)
    assert net.packed_weights() is pw1

    class Parent(torch.nn.Module):
        def __init__(self, rep):
            super().__init__()
            self.representation = rep

    parent, donor = Parent(net), Parent(mk())
    parent.load_state_dict(donor.state_dict(), assign=True)                     # never reaches GotenNet.load_state_dict
    pw2 = net.packed_weights()
    assert pw2 is not pw1 and torch.equal(pw2.layers[0].Wn1[:32], donor.representation.gata_list[0].W_q.weight)
    # functional_call: the swapped-in tensors are packed inside the call, the module's own again after it
    other = {k: v.detach().clone() + 1.0 for k, v in net.named_parameters()}
    seen = {}

    def probe(self_net):
        seen["inside"] = self_net.packed_weights().layers[0].Wn1[:32].clone()
        return torch.zeros(())

    orig_forward = type(net).forward
    try:
        type(net).forward = lambda self, *a, **k: probe(self)
        torch.func.functional_call(net, other, ())
    finally:
        type(net).forward = orig_forward
    assert torch.equal(seen["inside"], other["gata_list.0.W_q.weight"])
    assert torch.equal(net.packed_weights().layers[0].Wn1[:32], net.gata_list[0].W_q.weight)


def test_packed_cache_sees_replaced_submodules():
    """``net.gata_list[i] = new_layer`` (or a swapped node_init) leaves the OLD module's parameter dicts intact,
    so an identity check of parameter slots alone keeps the stale pack forever.  The submodule slots are part of the key."""
    import gotennet_amd
    mk = lambda: gotennet_amd.GotenNet(n_atom_basis=32, n_interactions=2, n_rbf=8, cutoff_fn=gotennet_amd.CosineCutoff(5.0), lmax=2)
    net, donor = mk(), mk()
    pw0 = net.packed_weights()
    assert net.packed_is_current()
    net.gata_list[0] = donor.gata_list[0]
    assert not net.packed_is_current()
    pw1 = net.packed_weights()
    assert pw1 is not pw0 and torch.equal(pw1.layers[0].Wn1[:32], donor.gata_list[0].W_q.weight)
    assert net.packed_is_current() and net.packed_weights() is pw1
    net.node_init = donor.node_init
    pw2 = net.packed_weights()
    assert pw2 is not pw1 and torch.equal(pw2.A_nbr, donor.node_init.A_nbr.weight)
    with torch.no_grad():
        net.eqff_list[1].W_vu.weight.add_(1.0)
    assert not net.packed_is_current()


# --------------------------------------------------------------------------------------------------- GPU
def _mirror(cfg, sd):
    from tests.test_hip_parity import _net_from_case
    return _net_from_case(cfg, sd)


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
@pytest.mark.parametrize("name", ["l2_sep_f32", "l3_sep_scale_f32", "opt_layernorm_tln", "opt_gauss_jointhtr_gated",
                                  "opt_mlp_linwa_ln_gated"])
def test_gata_and_eqff_modules_are_callable(name):
    """gotennet_amd.GATA / EQFF used the way a caller composing layers uses the reference's (gotennet.py:995-1007):
    every layer's (h, X, t) against the oracle's per-layer functions; also with a shuffled edge list."""
    from oracle import gotennet_oracle as orc
    cfg, sd, _, t = load_case(name)
    net = _mirror(cfg, sd)
    ei, ed = t["edge_index"], t["edge_diff"]
    N, E = t["z"].shape[0], ei.shape[1]
    # layer inputs from the oracle (pinned to the reference by the golden tests)
    _, _, tr = orc.gotennet_forward(sd, cfg, t["z"], ei, ed, t["edge_vec"], return_trace=True)
    rl = tr["rl"]
    deg = torch.zeros(N).index_add_(0, ei[0], torch.ones(E))
    n_edges = deg[ei[0]]
    L = cfg["n_interactions"]
    g = torch.Generator().manual_seed(0)
    perm = torch.randperm(E, generator=g)
    for li in range(L - 1):
        h_in, X_in, t_in = tr["layers"][li]                      # outputs of layer li = inputs of layer li + 1
        p, pe = f"gata_list.{li + 1}.", f"eqff_list.{li + 1}."
        hn, Xn = orc.gata_input_norms(sd, cfg, p, h_in, X_in)
        h_ref, X_ref = orc.gata_message_aggregate(sd, cfg, p, ei, hn, Xn, rl, t_in, ed, n_edges)
        t_ref = t_in
        if li + 1 != L - 1 and cfg.get("edge_updates", True):
            t_ref = orc.gata_htr(sd, cfg, p, ei, X_ref, rl, t_in)
        gata, eqff = net.gata_list[li + 1], net.eqff_list[li + 1]
        for order in (None, perm):
            sel = (lambda v: v) if order is None else (lambda v: v[order])
            eic = (ei if order is None else ei[:, order]).cuda()
            h1, X1, t1 = gata(eic, h_in.unsqueeze(1).cuda(), X_in.cuda(), sel(rl).cuda(), sel(t_in).cuda(), sel(ed).cuda(),
                              sel(n_edges).unsqueeze(1).cuda())
            assert h1.shape == (N, 1, cfg["n_atom_basis"])
            assert rel_err(h1.squeeze(1).cpu(), h_ref) < TOL and rel_err(X1.cpu(), X_ref) < TOL
            assert rel_err(t1.cpu(), sel(t_ref)) < TOL
        h2_ref, X2_ref = orc.eqff(sd, cfg, pe, h_ref, X_ref)
        h2, X2 = eqff(h1, X1)
        assert rel_err(h2.squeeze(1).cpu(), h2_ref) < TOL and rel_err(X2.cpu(), X2_ref) < TOL


@pytest.mark.gpu
def test_energy_forces_accepts_any_edge_order_and_rejects_bad_indices():
    """EnergyForces / CapturedStep validate the caller's edge list on the device: a shuffled list gives the sorted
    list's energies and forces bit for bit (stable sort: same per-target order), out-of-range indices raise."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.pipeline import CapturedStep, EnergyForces
    cfg, sd, head_sd, t = load_case("l2_sep_f32")
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    ef = EnergyForces(net, head)
    z, batch = t["z"].cuda(), t["batch"].cuda()
    ei, ed, ev = t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda()
    e0, f0 = (v.clone() for v in ef(z, ei, ed, ev, batch, cfg["n_mol"]))
    assert rel_err(f0.cpu(), t["forces"]) < TOL
    # permute whole target rows (keeps the order inside a row, so the stable sort restores the original list)
    E = ei.shape[1]
    tgt = ei[1]
    key = (tgt * 7919 + 13) % 1009                              # scrambles the row order deterministically
    order = torch.sort(key, stable=True).indices
    assert not bool((tgt[order][1:] >= tgt[order][:-1]).all())
    e1, f1 = ef(z, ei[:, order], ed[order], ev[order], batch, cfg["n_mol"])
    assert torch.equal(e1, e0) and torch.equal(f1, f0)
    # a fully random order: same sums in a different order inside the rows -> equal to rounding
    g = torch.Generator().manual_seed(1)
    rnd = torch.randperm(E, generator=g).cuda()
    e2, f2 = ef(z, ei[:, rnd], ed[rnd], ev[rnd], batch, cfg["n_mol"])
    assert rel_err(e2.cpu(), e0.cpu()) < 1e-5 and rel_err(f2.cpu(), f0.cpu()) < 1e-4
    bad = ei.clone()
    bad[0, 3] = z.shape[0]                                       # one past the last atom
    with pytest.raises(ValueError):
        ef(z, bad, ed, ev, batch, cfg["n_mol"])
    with pytest.raises(ValueError):
        CapturedStep(ef, z, bad, batch, cfg["n_mol"])
    with pytest.raises(ValueError):
        net(z, bad, ed, ev)
    if cfg["n_mol"] > 1:                                         # an unsorted batch vector is rejected, not mis-summed
        with pytest.raises(ValueError):
            ef(z, ei, ed, ev, batch.flip(0).contiguous(), cfg["n_mol"])
    # the molecule-offset kernel stays in bounds on ANY batch vector (negative / too large / unsorted entries)
    from gotennet_amd.outputs import molecule_ptr
    for bv in ([3, -2, 9, 0, 1], [5, 5, 5], [-1, -1], [0, 2, 1, 2, 0]):
        mp_ = molecule_ptr(torch.tensor(bv, dtype=torch.int64).cuda(), 3).cpu()
        assert mp_.shape == (4,) and int(mp_.min()) >= 0 and int(mp_.max()) <= len(bv)
    # CapturedStep on a shuffled list: same result as the eager path
    step = CapturedStep(ef, z, ei[:, order], batch, cfg["n_mol"])
    e3, f3 = step(t["pos"].cuda())
    assert rel_err(e3.cpu(), e0.cpu()) < 1e-5 and rel_err(f3.cpu(), f0.cpu()) < 1e-4


@pytest.mark.gpu
def test_head_scale_shift_atomref_against_reference():
    """gn_head_energy with non-trivial mean / stddev / atomref against the reference Atomwise (tests/golden/kat_head.npz),
    and the force path: F scales with stddev, E = stddev E0 + n mean + sum atomref[z]."""
    from gotennet_amd.outputs import Atomwise, molecule_ptr
    t, hsd = _head_kat()
    F_, Hd = t["h"].shape[1], hsd["out_net.1.out_net.0.weight"].shape[0]
    head = Atomwise(n_in=F_, n_hidden=Hd, property="property", contributions="contrib", mean=hsd["standardize.mean"],
                    stddev=hsd["standardize.stddev"], atomref=t["atomref"], activation="silu")
    head.load_state_dict(hsd, strict=True)
    head = head.cuda().eval()
    n_mol = int(t["n_mol"])
    z32, batch = t["z"].cuda().to(torch.int32), t["batch"].cuda()
    e, y, _ = head.energy_raw(t["h"].cuda(), z32, molecule_ptr(batch, n_mol), n_mol)
    assert rel_err(e.cpu(), t["energy"]) < 1e-5
    assert rel_err(y.cpu(), t["contrib"].reshape(-1)) < 1e-5
    # reference-style call
    class _In(dict):
        __getattr__ = dict.__getitem__
    out = head(_In(z=t["z"].cuda(), batch=batch, pos=None, representation=t["h"].cuda()))
    assert rel_err(out["property"].cpu(), t["energy"]) < 1e-5 and rel_err(out["contrib"].cpu(), t["contrib"]) < 1e-5
    # through the fused force pipeline
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, c = load_case("l2_sep_f32")
    net = _mirror(cfg, sd)
    plain = _head_from_case(cfg, head_sd)
    atomref = torch.linspace(-2.0, 3.0, cfg["max_z"]).reshape(-1, 1)
    scaled = Atomwise(n_in=cfg["n_atom_basis"], n_hidden=16, property="property", derivative="forces",
                      mean=torch.tensor([1.7]), stddev=torch.tensor([0.35]), atomref=atomref, activation="silu")
    scaled.load_state_dict({**head_sd, "standardize.mean": torch.tensor([1.7]), "standardize.stddev": torch.tensor([0.35]),
                            "atomref.weight": atomref}, strict=True)
    scaled = scaled.cuda().eval()
    args = (c["z"].cuda(), c["edge_index"].cuda(), c["edge_diff"].cuda(), c["edge_vec"].cuda(), c["batch"].cuda(), cfg["n_mol"])
    e0, f0 = (v.clone() for v in EnergyForces(net, plain)(*args))
    e1, f1 = EnergyForces(net, scaled)(*args)
    cnt = torch.bincount(c["batch"], minlength=cfg["n_mol"]).double()
    ref_sum = torch.zeros(cfg["n_mol"], dtype=torch.double).index_add_(0, c["batch"], atomref.double()[c["z"], 0])
    e_expect = 0.35 * e0.cpu().double().reshape(-1) + 1.7 * cnt + ref_sum
    assert rel_err(e1.cpu().reshape(-1), e_expect) < 1e-5
    assert rel_err(f1.cpu(), 0.35 * f0.cpu()) < 1e-5


@pytest.mark.gpu
def test_training_mode_warns_once_and_stale_pack_is_refreshed():
    import gotennet_amd
    cfg, sd, _, t = load_case("l2_sep_f32")
    net = _mirror(cfg, sd)
    args = (t["z"].cuda(), t["edge_index"].cuda(), t["edge_diff"].cuda())
    h0, _ = net(*args, t["edge_vec"].cuda())
    with torch.no_grad():
        net.gata_list[0].W_q.weight.mul_(1.5)
    h1, _ = net(*args, t["edge_vec"].cuda())
    assert not torch.equal(h0, h1)                               # the packed copy followed the in-place update
    net.train()
    ev = t["edge_vec"].cuda().requires_grad_(True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        net(*args, ev)
        net(*args, ev)
    assert sum("inference" in str(x.message) for x in w) == 1


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
def test_deep_head_mean_aggregation_against_reference():
    """Atomwise(n_layers=3, aggregation_mode="mean") with the reference's default activation on the HIP path against the
    reference's own head (tests/golden/kat_head_l3_mean_ssp.npz): per-atom contributions and per-molecule means."""
    from gotennet_amd.outputs import Atomwise, molecule_ptr
    t, hsd = _head_kat3()
    head = Atomwise(n_in=t["h"].shape[1], n_layers=3, aggregation_mode="mean", property="property", contributions="contrib",
                    mean=hsd["standardize.mean"], stddev=hsd["standardize.stddev"])
    head.load_state_dict(hsd, strict=True)
    head = head.cuda().eval()
    n_mol = int(t["n_mol"])
    e, y, _ = head.energy_raw(t["h"].cuda(), t["z"].cuda().to(torch.int32), molecule_ptr(t["batch"].cuda(), n_mol), n_mol)
    assert rel_err(y.cpu(), t["contrib"].reshape(-1)) < 1e-5
    assert rel_err(e.cpu(), t["energy"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n_layers,n_hidden,agg,act", [(1, None, "sum", "silu"), (3, None, "mean", "softplus"),
                                                     (3, [48, 24], "sum", "tanh"), (2, 40, None, "gelu"),
                                                     (4, 32, "mean", "elu")])
def test_head_shapes_energy_and_forces_match_oracle(n_layers, n_hidden, agg, act):
    """Every SchnetMLP shape / aggregation the head accepts: energies (or per-atom values) and forces through the fused
    pipeline and through the reference-style autograd call, against the oracle's autograd."""
    import types
    import gotennet_amd
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    from oracle import gotennet_oracle as orc
    cfg, sd, _, t = load_case("l2_sep_f32")
    F = cfg["n_atom_basis"]
    torch.manual_seed(n_layers * 7 + len(act))
    head = Atomwise(n_in=F, n_layers=n_layers, n_hidden=n_hidden, aggregation_mode=agg, activation=act, property="y",
                    derivative="forces", mean=torch.tensor([0.3]), stddev=torch.tensor([1.7]))
    with torch.no_grad():
        for p in head.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    e_ref, f_ref, _ = orc.energy_and_forces({k: v.double() for k, v in sd.items()}, cfg, {k: v.double() for k, v in hsd.items()},
                                            t["z"], t["pos"].double(), t["batch"], cfg["n_mol"], activation=act,
                                            aggregation="mean" if agg == "mean" else "sum")
    net = gotennet_amd.GotenNetWrapper(
        n_atom_basis=F, n_interactions=cfg["n_interactions"], n_rbf=cfg["n_rbf"], cutoff_fn=gotennet_amd.CosineCutoff(cfg["cutoff"]),
        max_z=cfg["max_z"], num_heads=cfg["num_heads"], scale_edge=cfg["scale_edge"], lmax=cfg["lmax"],
        sep_dir=cfg["sep_dir"], sep_tensor=cfg["sep_tensor"])
    net.load_state_dict(sd, strict=True)
    net, head = net.cuda().eval(), head.cuda().eval()
    e, f = EnergyForces(net, head)(t["z"].cuda(), t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda(),
                                   t["batch"].cuda(), cfg["n_mol"])
    assert rel_err(e.cpu(), e_ref) < TOL and rel_err(f.cpu(), f_ref) < TOL
    # reference-style: representation + head.forward with autograd.grad inside
    pos = t["pos"].cuda().requires_grad_(True)
    inp = types.SimpleNamespace(z=t["z"].cuda(), pos=pos, batch=t["batch"].cuda())
    inp.representation, inp.vector_representation = net(inp)
    out = head(inp)
    if agg is None:                                  # per-atom values, no scatter (outputs.py:354-358)
        y_ref = orc.atomwise_contributions({k: v.double() for k, v in hsd.items()},
                                           orc.gotennet_forward({k: v.double() for k, v in sd.items()}, cfg, t["z"], t["edge_index"],
                                                                t["edge_diff"].double(), t["edge_vec"].double())[0],
                                           t["z"], activation=act)
        assert out["y"].shape == (t["z"].shape[0], 1) and rel_err(out["y"].detach().cpu(), y_ref) < TOL
    else:
        assert rel_err(out["y"].detach().cpu(), e_ref) < TOL
    assert rel_err(out["forces"].detach().cpu(), f_ref) < TOL


# ---------------------------------------------------------------- vector read-outs of the QM9 task (Dipole, ESE)
def _qm9_kat(dtype=torch.float32):
    k = np.load(os.path.join(GOLDEN_DIR, "kat_qm9_heads.npz"))
    t = {n: torch.from_numpy(k[n]) for n in k.files if "/" not in n}
    sd = {tag: {n[len(tag) + 1:]: torch.from_numpy(k[n]) for n in k.files if n.startswith(tag + "/")}
          for tag in ("dip_task", "dip_vec", "ese")}
    return t, sd


def test_oracle_qm9_heads_match_reference_kat():
    """Dipole as QM9Task builds it (magnitude, standardised charges), Dipole in vector form with n_hidden != n_in, and
    ElectronicSpatialExtentV2, against the reference's own outputs (tools/make_golden.py qm9_heads_kat)."""
    from oracle import gotennet_oracle as orc
    t, sd = _qm9_kat()
    h, X, pos, z, batch, n_mol = t["h"], t["X"], t["pos"], t["z"], t["batch"], int(t["n_mol"])
    y, yv = orc.dipole(sd["dip_task"], h, X, pos, batch, n_mol, "silu", mean=torch.tensor(0.3), stddev=torch.tensor(1.7),
                       predict_magnitude=True)
    assert y.shape == t["dip_task_y"].shape == (n_mol, 1) and yv.shape == t["dip_task_yvec"].shape == (n_mol, 3, 1)
    assert rel_err(y, t["dip_task_y"]) < 2e-6 and rel_err(yv, t["dip_task_yvec"]) < 2e-6
    y, yv = orc.dipole(sd["dip_vec"], h, X, pos, batch, n_mol, "silu")
    assert y.shape == (n_mol, 3)
    assert rel_err(y, t["dip_vec_y"]) < 2e-6 and rel_err(yv, t["dip_vec_yvec"]) < 2e-6
    y, x = orc.electronic_spatial_extent(sd["ese"], h, pos, z, batch, n_mol, "softplus")
    assert rel_err(y, t["ese_y"]) < 2e-6 and rel_err(x, t["ese_contrib"]) < 2e-6
    assert torch.equal(sd["ese"]["atomic_mass"], t["masses"])


def test_qm9_head_state_dict_layout():
    import gotennet_amd.outputs as out
    _, sd = _qm9_kat()
    dt = out.Dipole(n_in=64, predict_magnitude=True, property="property", mean=torch.tensor(0.3), stddev=torch.tensor(1.7))
    dv = out.Dipole(n_in=64, n_hidden=32, property="dipole")
    es = out.ElectronicSpatialExtentV2(n_in=64, property="property", contributions="contrib")
    for mod, tag in ((dt, "dip_task"), (dv, "dip_vec"), (es, "ese")):
        assert sorted(mod.state_dict().keys()) == sorted(sd[tag].keys()), tag
        mod.load_state_dict(sd[tag], strict=True)
    assert float(es.atomic_mass[6]) == pytest.approx(12.011) and float(es.atomic_mass[0]) == 1.0


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
def test_qm9_heads_against_reference():
    """The HIP Dipole / ElectronicSpatialExtentV2 modules on the reference's KAT: inputs as the QM9 task hands them over
    (duck-typed batch with .representation / .vector_representation [N, D, F], the block takes the X[:, :3] view)."""
    import types
    import gotennet_amd.outputs as out
    t, sd = _qm9_kat()
    n_mol = int(t["n_mol"])
    inp = types.SimpleNamespace(z=t["z"].cuda(), batch=t["batch"].cuda(), pos=t["pos"].cuda(),
                                representation=t["h"].cuda(), vector_representation=t["X"].cuda())
    dt = out.Dipole(n_in=64, predict_magnitude=True, property="property", mean=torch.tensor(0.3), stddev=torch.tensor(1.7))
    dt.load_state_dict(sd["dip_task"], strict=True)
    r = dt.cuda().eval()(inp)
    assert r["property"].shape == (n_mol, 1) and r["property_vector"].shape == (n_mol, 3, 1)
    assert rel_err(r["property"].cpu(), t["dip_task_y"]) < 1e-4
    assert rel_err(r["property_vector"].cpu(), t["dip_task_yvec"]) < 1e-4
    dv = out.Dipole(n_in=64, n_hidden=32, property="dipole")
    dv.load_state_dict(sd["dip_vec"], strict=True)
    r = dv.cuda().eval()(inp)
    assert r["dipole"].shape == (n_mol, 3)
    assert rel_err(r["dipole"].cpu(), t["dip_vec_y"]) < 1e-4 and rel_err(r["dipole_vector"].cpu(), t["dip_vec_yvec"]) < 1e-4
    es = out.ElectronicSpatialExtentV2(n_in=64, property="property", contributions="contrib")
    es.load_state_dict(sd["ese"], strict=True)
    r = es.cuda().eval()(inp)
    assert rel_err(r["property"].cpu(), t["ese_y"]) < 1e-4 and rel_err(r["contrib"].cpu(), t["ese_contrib"]) < 1e-4
    # a standalone block on a contiguous [N, 3, F] input gives the same as on the X[:, :3] view
    blk = dv.equivariant_layers[0]
    s1, v1 = blk(inp.representation, inp.vector_representation[:, :3, :])
    s2, v2 = blk(inp.representation, inp.vector_representation[:, :3, :].contiguous())
    assert torch.equal(s1, s2) and torch.equal(v1, v2)


@pytest.mark.gpu
def test_qm9_heads_on_the_representation_match_oracle():
    """End to end: GotenNetWrapper -> Dipole / ESE on a 2-molecule batch against the oracle (F = 32, lmax = 2)."""
    import types
    import gotennet_amd
    import gotennet_amd.outputs as out
    from oracle import gotennet_oracle as orc
    from tests.test_hip_parity import _synthetic
    torch.manual_seed(7)
    kw = dict(n_atom_basis=32, n_interactions=2, n_rbf=8, num_heads=8, scale_edge=False, lmax=2, sep_dir=True, sep_tensor=True)
    net = gotennet_amd.GotenNetWrapper(cutoff_fn=gotennet_amd.CosineCutoff(5.0), **kw)
    dip = out.Dipole(n_in=32, predict_magnitude=True, property="mu")
    ese = out.ElectronicSpatialExtentV2(n_in=32, property="r2")
    sd = {k: v.clone().double() for k, v in net.state_dict().items()}
    dsd = {k: v.clone().double() for k, v in dip.state_dict().items()}
    esd = {k: v.clone().double() for k, v in ese.state_dict().items()}
    pos, batch, z = _synthetic(2, 9, 3.0, seed=3)
    z = z.clamp(max=9)
    cfg = orc.default_config(**kw)
    ei, w, vec = orc.distance(pos.double(), batch, 5.0)
    h, X = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
    y_mu, _ = orc.dipole(dsd, h, X, pos.double(), batch, 2, "silu", predict_magnitude=True)
    y_r2, _ = orc.electronic_spatial_extent(esd, h, pos.double(), z, batch, 2, "softplus")
    net, dip, ese = net.cuda().eval(), dip.cuda().eval(), ese.cuda().eval()
    inp = types.SimpleNamespace(z=z.cuda(), pos=pos.cuda(), batch=batch.cuda())
    with torch.no_grad():
        inp.representation, inp.vector_representation = net(inp)
        assert rel_err(dip(inp)["mu"].cpu(), y_mu) < 1e-4
        assert rel_err(ese(inp)["r2"].cpu(), y_r2) < 1e-4


@pytest.mark.gpu
def test_energy_forces_topology_cache():
    """EnergyForces keeps the CSR / CSC arrays of the last edge list: a second call with the SAME edge_index tensor and
    new geometry gives exactly what a cache-less instance gives; an in-place edit of the tensor (version counter) or
    another tensor rebuilds; a shuffled list keeps its sort permutation across the cached call."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case("l2_sep_f32")
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    z, batch = t["z"].cuda(), t["batch"].cuda()
    ei, ed, ev = t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda()
    cached, plain = EnergyForces(net, head), EnergyForces(net, head, cache_topology=False)
    e0, f0 = (v.clone() for v in cached(z, ei, ed, ev, batch, cfg["n_mol"]))
    g_first = cached._topo[2]
    # new geometry on the same list: scaled vectors (self-loops stay at zero length)
    ev2, ed2 = ev * 0.97, ed * 0.97
    e1, f1 = (v.clone() for v in cached(z, ei, ed2, ev2, batch, cfg["n_mol"]))
    assert cached._topo[2] is g_first                                    # topology reused
    e1p, f1p = plain(z, ei, ed2, ev2, batch, cfg["n_mol"])
    assert torch.equal(e1, e1p) and torch.equal(f1, f1p)
    assert not torch.equal(e1, e0)
    # forward-only call on the cached topology, then forces again
    e2, none = cached(z, ei, ed, ev, batch, cfg["n_mol"], forces=False)
    assert none is None and rel_err(e2.cpu(), e0.cpu()) < 1e-6
    e3, f3 = cached(z, ei, ed, ev, batch, cfg["n_mol"])
    assert torch.equal(e3, e0) and torch.equal(f3, f0)
    # a shuffled list: the cached call re-applies the stored permutation to the new geometry
    E = ei.shape[1]
    order = torch.randperm(E, generator=torch.Generator().manual_seed(3)).cuda()
    eis = ei[:, order].contiguous()
    e4, f4 = (v.clone() for v in cached(z, eis, ed[order], ev[order], batch, cfg["n_mol"]))
    g_shuf = cached._topo[2]
    e5, f5 = cached(z, eis, ed2[order], ev2[order], batch, cfg["n_mol"])
    assert cached._topo[2] is g_shuf
    assert rel_err(f4.cpu(), f0.cpu()) < 1e-5 and rel_err(f5.cpu(), f1.cpu()) < 1e-5
    # an in-place edit of the edge list is seen (version counter): dropping the last edge's twin changes the result
    eis.add_(0)
    cached(z, eis, ed[order], ev[order], batch, cfg["n_mol"])
    assert cached._topo[2] is not g_shuf


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
@pytest.mark.parametrize("case", ["l1_nosep_scale_f32", "l2_sep_f32", "l2_mixed_f64ch", "c2_model_3mol_seeded",
                                  "l3_sep_scale_f32", "l4_sep_f32", "c2_model_lmax4_1mol_seeded"])
def test_first_interaction_without_tensor_gate_blocks(case):
    """GotenNet.forward starts from X = 0 (gotennet.py:992), so the first interaction's tensor-gate terms are 0 * gate:
    the engine skips those blocks of the edge projection (forward N-prefix, backward K-prefix), of x / v, and the X_in
    gathers (X_in = NULL forms of gn_message_aggregate / gn_message_backward; at lmax 3-4 one launch instead of the degree
    groups).  Same energies and forces as the general
    kernels run on the zero tensor: bit-identical where the GEMM tile shape does not change with the narrower product,
    else at the arithmetic's own batch-layout noise (<= 2e-6)."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd import engine
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case(case)
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    z, batch = t["z"].cuda(), t["batch"].cuda()
    ei, ed, ev = t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda()
    out = {}
    for flag in (True, False):
        old = engine.ZERO_X_FIRST
        engine.ZERO_X_FIRST = flag
        try:
            e, f = EnergyForces(net, head, cache_topology=False)(z, ei, ed, ev, batch, cfg["n_mol"])
            h, X = net(z, ei, ed, ev)
            out[flag] = (e.cpu(), f.cpu(), h.detach().cpu(), X.detach().cpu())
        finally:
            engine.ZERO_X_FIRST = old
    for a, b in zip(out[True], out[False]):
        assert rel_err(a, b) < 2e-6
    if engine.GEMM_MODE != "f16x2":                  # row-wise arithmetics: the narrower products give the same bits
        e1, f1, h1, X1 = out[True]
        e0, f0, h0, X0 = out[False]
        assert torch.equal(e1, e0) and torch.equal(h1, h0) and torch.equal(X1, X0)
        if cfg["lmax"] <= 2:                         # above, the general backward sums its g_cut per degree group
            assert torch.equal(f1, f0)


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
@pytest.mark.parametrize("case", ["l1_nosep_scale_f32", "l2_sep_f32", "l2_mixed_f64ch", "c2_model_3mol_seeded", "l4_sep_f32"])
def test_message_backward_merged_kernel_vs_kernel_pair(case):
    """gn_message_backward with a head-sum workspace runs the merged by-source kernel (own rows in LDS, its first-interaction
    form included); without one, the by-target / by-source pair (what direct callers of the C entry get when they pass
    ga_parts = NULL).  Same gradients: forces agree at the level of a changed summation order, both match the reference."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd import engine
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case(case)
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    z, batch = t["z"].cuda(), t["batch"].cuda()
    ei, ed, ev = t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda()
    out = {}
    for pair in (False, True):
        engine.MSG_BWD_PAIR = pair
        try:
            e, f = EnergyForces(net, head, cache_topology=False)(z, ei, ed, ev, batch, cfg["n_mol"])
            out[pair] = (e.cpu(), f.cpu())
        finally:
            engine.MSG_BWD_PAIR = False
    assert torch.equal(out[False][0], out[True][0])
    assert rel_err(out[True][1], out[False][1]) < 2e-6
    assert rel_err(out[True][1], t["forces"]) < 1e-4 and rel_err(out[False][1], t["forces"]) < 1e-4


@pytest.mark.gpu
def test_gata_module_edge_cases():
    """GATA.forward: no edges -> the (normalised) inputs come back; an n_edges that is not the out-degree of the
    sources raises instead of being ignored."""
    cfg, sd, _, t = load_case("l3_sep_scale_f32")                      # scale_edge=True: n_edges matters
    net = _mirror(cfg, sd)
    gata = net.gata_list[0]
    N, F_ = t["z"].shape[0], cfg["n_atom_basis"]
    D = (cfg["lmax"] + 1) ** 2 - 1
    g = torch.Generator().manual_seed(0)
    h, X = torch.randn(N, 1, F_, generator=g).cuda(), torch.randn(N, D, F_, generator=g).cuda()
    empty = torch.zeros((2, 0), dtype=torch.long, device="cuda")
    h1, X1, t1 = gata(empty, h, X, torch.zeros(0, D, device="cuda"), torch.zeros(0, F_, device="cuda"),
                      torch.zeros(0, device="cuda"))
    assert torch.equal(h1, h) and torch.equal(X1, X) and t1.shape == (0, F_)
    ei, ed = t["edge_index"].cuda(), t["edge_diff"].cuda()
    E = ei.shape[1]
    from oracle import gotennet_oracle as orc
    _, _, tr = orc.gotennet_forward(sd, cfg, t["z"], t["edge_index"], t["edge_diff"], t["edge_vec"], return_trace=True)
    rl, tij = tr["rl"].cuda(), torch.randn(E, F_, generator=g).cuda()
    if cfg["scale_edge"]:
        with pytest.raises(ValueError):
            gata(ei, h, X, rl, tij, ed, torch.full((E, 1), 3.0, device="cuda"))


@pytest.mark.gpu
def test_two_models_two_arithmetics_two_threads():
    """The projection arithmetic and the activation kind are per-model values carried in ``Config`` (no module-level
    call state): an exact-fp32 model and a 2xfp16-split model (with different activations) run CONCURRENTLY from two
    threads, each on its own stream, and every result is bit-identical to the same model run alone."""
    import threading
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.pipeline import EnergyForces
    jobs = []
    for case, mode in (("l2_sep_f32", "f32"), ("opt_act_tanh_l3_gated", "f16x2")):
        cfg, sd, head_sd, t = load_case(case)
        net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
        net.gemm_mode = mode
        assert net.config().gemm_mode == mode
        args = [t[k].cuda() for k in ("z", "edge_index", "edge_diff", "edge_vec", "batch")] + [cfg["n_mol"]]
        ef = EnergyForces(net, head)
        e0, f0 = ef(*args)
        torch.cuda.synchronize()
        assert rel_err(e0.cpu(), t["energy"]) < TOL and rel_err(f0.cpu(), t["forces"]) < TOL
        jobs.append((ef, args, e0.clone(), f0.clone()))
    bad, barrier = [], threading.Barrier(2)

    def run(ef, args, e0, f0):
        stream = torch.cuda.Stream()
        barrier.wait()
        with torch.cuda.stream(stream):
            for _ in range(30):
                e, f = ef(*args)
                if not (torch.equal(e, e0) and torch.equal(f, f0)):
                    bad.append(ef.rep.gemm_mode)
        stream.synchronize()

    ths = [threading.Thread(target=run, args=j) for j in jobs]
    [th.start() for th in ths]
    [th.join() for th in ths]
    assert not bad, bad


@pytest.mark.gpu
def test_energy_forces_under_inference_mode():
    """A neighbour list built under torch.inference_mode() has no version counter (reading ``_version`` raises): the
    topology cache must treat it as a miss, not crash (the round-3 code did)."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case("l2_sep_f32")
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    ef = EnergyForces(net, head)
    z, batch = t["z"].cuda(), t["batch"].cuda()
    e0, f0 = ef(z, t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda(), batch, cfg["n_mol"])
    with torch.inference_mode():
        ei, ed, ev = t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda()
        assert ei.is_inference()
        for _ in range(2):
            e, f = ef(z, ei, ed, ev, batch, cfg["n_mol"])
    assert torch.equal(e, e0) and torch.equal(f, f0)
    assert ef._topo is None                          # never cached
    ef.clear_cache()


@pytest.mark.gpu
def test_energy_forces_replays_static_topology_bit_identically():
    """EnergyForces(replay=True): after two calls on one topology the step is recorded into one hipGraph and replayed on
    the call's fresh edge_diff / edge_vec / z -- bit-identical to the eager step for NEW positions, an eager call when the
    edge list changes, and again a capture on the new list."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case("l2_sep_f32")
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    z, batch, pos = t["z"].cuda(), t["batch"].cuda(), t["pos"].cuda()
    eager, rep = EnergyForces(net, head), EnergyForces(net, head, replay=True)
    ei, ed, ev = distance(pos, batch, cfg["cutoff"], 32)
    g = torch.Generator().manual_seed(3)
    for it in range(6):
        p2 = pos + 1e-3 * torch.randn(pos.shape, generator=g).cuda()       # same neighbour list, new geometry
        ei2, ed2, ev2 = distance(p2, batch, cfg["cutoff"], 32)
        assert torch.equal(ei2, ei)
        e0, f0 = eager(z, ei, ed2, ev2, batch, cfg["n_mol"])
        e1, f1 = rep(z, ei, ed2, ev2, batch, cfg["n_mol"])
        assert torch.equal(e0, e1) and torch.equal(f0, f1), it
        assert (rep._graph_state is not None) == (it >= 2)
    # another edge list (one edge dropped): eager again, same numbers as the eager object, then a new capture
    keep = torch.ones(ei.shape[1], dtype=torch.bool, device="cuda")
    keep[int((ei[0] != ei[1]).nonzero()[0])] = False
    ei3, ed3, ev3 = ei[:, keep].contiguous(), ed[keep].contiguous(), ev[keep].contiguous()
    for it in range(4):
        e0, f0 = eager(z, ei3, ed3, ev3, batch, cfg["n_mol"])
        e1, f1 = rep(z, ei3, ed3, ev3, batch, cfg["n_mol"])
        assert torch.equal(e0, e1) and torch.equal(f0, f1)
    assert rep._graph_state is not None
    # a caller-ordered (shuffled) edge list: the cached stable-sort permutation is applied to every call's edge_diff / edge_vec
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(5)).cuda()
    ei4 = ei[:, perm].contiguous()
    for it in range(4):
        p2 = pos + 1e-3 * torch.randn(pos.shape, generator=g).cuda()
        _, ed2, ev2 = distance(p2, batch, cfg["cutoff"], 32)
        ed4, ev4 = ed2[perm].contiguous(), ev2[perm].contiguous()
        e0, f0 = eager(z, ei4, ed4, ev4, batch, cfg["n_mol"])                     # the same shuffled list, eager
        e1, f1 = rep(z, ei4, ed4, ev4, batch, cfg["n_mol"])
        assert torch.equal(e0, e1) and torch.equal(f0, f1), it
        et, ft = eager(z, ei, ed2, ev2, batch, cfg["n_mol"])                      # target-major list: another order inside a
        assert rel_err(e1, et) < 1e-5 and rel_err(f1, ft) < 1e-5                  # target's row, fp32 re-association only
    assert rep._graph_state is not None and rep._graph_state["order"] is not None
    rep.clear_cache()
    assert rep._graph_state is None and rep._topo is None


@pytest.mark.gpu
def test_replay_follows_the_head():
    """The recorded step bakes the head's host scalars (last bias, scale / shift) into kernel arguments and captures the
    pointers of its transposed weights: a head that changes after the capture -- an in-place update, load_state_dict, a
    new mean / stddev -- must drop the graph, not replay the old head."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case("l2_sep_f32")
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    z, batch, pos = t["z"].cuda(), t["batch"].cuda(), t["pos"].cuda()
    ei, ed, ev = distance(pos, batch, cfg["cutoff"], 32)
    rep = EnergyForces(net, head, replay=True)
    for _ in range(4):
        e_old, f_old = rep(z, ei, ed, ev, batch, cfg["n_mol"])
    assert rep._graph_state is not None
    with torch.no_grad():                                                       # optimiser-style in-place update of the head
        for p_ in head.parameters():
            p_.mul_(1.5)
        if hasattr(head.standardize, "stddev"):
            head.standardize.stddev.mul_(2.0)
            head.standardize.mean.add_(0.25)
    e_new, f_new = rep(z, ei, ed, ev, batch, cfg["n_mol"])
    e_ref, f_ref = EnergyForces(net, head)(z, ei, ed, ev, batch, cfg["n_mol"])
    assert torch.equal(e_new, e_ref) and torch.equal(f_new, f_ref)
    assert not torch.equal(e_new, e_old)
    for _ in range(3):                                                          # ... and records again on the new head
        e2, f2 = rep(z, ei, ed, ev, batch, cfg["n_mol"])
    assert rep._graph_state is not None and torch.equal(e2, e_ref) and torch.equal(f2, f_ref)


@pytest.mark.gpu
def test_bench_one_rank_rccl_path():
    """`bench.py --force-dist`: RCCL initialised with ONE rank (the boxes the driver reaches have one GPU): process group,
    barrier, the per-step all-reduce of the energy vector all run; the record says one rank was seen and its checksum
    equals the plain run's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-lmax4", "--no-split",
            "--no-graph", "--no-workloads", "--no-cpu-baseline", "--no-forward-only", "--no-live-traffic"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = []
    for extra in ([], ["--force-dist"]):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    plain, dist_ = outs
    assert dist_["n_ranks_seen"] == 1 and dist_["energy_vector_len"] == 8
    assert dist_["energy_checksum"] == plain["energy_checksum"]
    assert dist_["rank_ms_per_step"] and len(dist_["rank_ms_per_step"]) == 1


# --------------------------------------------------------------------------------------------------- AtomwiseV3
def _head_kat_v3():
    k = np.load(os.path.join(GOLDEN_DIR, "kat_head_v3.npz"))
    t = {n: torch.from_numpy(k[n]) for n in k.files if not n.startswith("head/")}
    hsd = {n[5:]: torch.from_numpy(k[n]) for n in k.files if n.startswith("head/")}
    return t, hsd


@pytest.mark.parametrize("agg", ["sum", "mean", "none"])
def test_oracle_atomwise_v3_matches_reference_kat(agg):
    """The oracle's AtomwiseV3 restatement (scale per atom, mean added after the aggregation; outputs.py:186-229) against
    the reference's own module, all three aggregation modes."""
    from oracle import gotennet_oracle as orc
    t, hsd = _head_kat_v3()
    y, yi = orc.atomwise_v3(hsd, t["h"], t["batch"], int(t["n_mol"]), float(t["mean"]), float(t["stddev"]), z=t["z"],
                            aggregation=None if agg == "none" else agg)
    assert rel_err(y, t[f"energy_{agg}"]) < 1e-6 and rel_err(yi, t[f"contrib_{agg}"]) < 1e-6


def test_atomwise_v3_state_dict_keys_match_reference():
    from gotennet_amd.outputs import AtomwiseV3
    t, hsd = _head_kat_v3()
    head = AtomwiseV3(n_in=64, n_hidden=32, activation="silu", mean=1.7, stddev=0.35, atomref=t["atomref"])
    assert set(head.state_dict().keys()) == set(hsd.keys())
    head.load_state_dict(hsd, strict=True)
    with pytest.raises(NotImplementedError):
        AtomwiseV3(n_in=64, n_out=3)


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
@pytest.mark.parametrize("agg", ["sum", "mean", "none"])
def test_atomwise_v3_matches_reference_kat(agg):
    """gotennet_amd.outputs.AtomwiseV3 through the reference-style call (``inputs.z/.batch/.pos/.representation``): property
    and contributions of the reference's AtomwiseV3, and d(property)/d(representation) against the oracle's autograd."""
    from gotennet_amd.outputs import AtomwiseV3
    from oracle import gotennet_oracle as orc
    t, hsd = _head_kat_v3()
    aggregation = None if agg == "none" else agg
    head = AtomwiseV3(n_in=64, n_hidden=32, activation="silu", property="property", contributions="contrib", mean=1.7,
                      stddev=0.35, atomref=t["atomref"], aggregation_mode=aggregation)
    head.load_state_dict(hsd, strict=True)
    head = head.cuda().eval()
    h = t["h"].cuda().requires_grad_(True)
    res = head(dict(z=t["z"].cuda(), batch=t["batch"].cuda(), pos=None, representation=h))
    assert rel_err(res["property"].detach().cpu(), t[f"energy_{agg}"]) < TOL
    assert rel_err(res["contrib"].detach().cpu(), t[f"contrib_{agg}"]) < TOL
    (gh,) = torch.autograd.grad(res["property"].sum(), h)
    h64 = t["h"].double().requires_grad_(True)
    y64, _ = orc.atomwise_v3({k: v.double() if v.is_floating_point() else v for k, v in hsd.items()}, h64, t["batch"],
                             int(t["n_mol"]), 1.7, 0.35, z=t["z"], aggregation=aggregation)
    (g64,) = torch.autograd.grad(y64.sum(), h64)
    assert rel_err(gh.cpu(), g64) < TOL


@pytest.mark.gpu
def test_device_csc_and_molecule_offsets_match_torch():
    """gn_build_csc (count / scan / scatter / per-bucket ranking) against a stable sort by source, bit-exact: random graphs
    with empty sources, a source of out-degree 700 / 3000, more atoms than one scan chunk, no edges; gn_molecule_ptr against
    bincount + cumsum incl. empty molecules at both ends."""
    from gotennet_amd._lib import call, ptr
    from gotennet_amd.outputs import molecule_ptr
    g = torch.Generator().manual_seed(0)
    for N, E in ((1, 1), (50, 0), (300, 4000), (5000, 90000), (3000, 2500)):
        src = torch.randint(0, N, (E,), generator=g)
        if E > 1000:
            src[:700 if E < 50000 else 3000] = 7                 # one hub (3000: three LDS windows of the ranking kernel)
            src[src == 11] = 12                                  # one source without edges
        dst = torch.sort(torch.randint(0, N, (E,), generator=g)).values       # target-major
        s32, d32 = src.to(torch.int32).cuda(), dst.to(torch.int32).cuda()
        colptr = torch.empty(N + 1, dtype=torch.int32, device="cuda")
        perm, tgt = torch.empty(E, dtype=torch.int32, device="cuda"), torch.empty(E, dtype=torch.int32, device="cuda")
        work = torch.empty(N + E, dtype=torch.int32, device="cuda")
        call("gn_build_csc", ptr(s32), ptr(d32), E, N, ptr(colptr), ptr(perm), ptr(tgt), ptr(work), None)
        torch.cuda.synchronize()
        ref_perm = torch.sort(src, stable=True).indices
        ref_col = torch.zeros(N + 1, dtype=torch.int64)
        ref_col[1:] = torch.bincount(src, minlength=N).cumsum(0)
        assert torch.equal(colptr.cpu().long(), ref_col)
        assert torch.equal(perm.cpu().long(), ref_perm) and torch.equal(tgt.cpu().long(), dst[ref_perm])
    for n_mol, sizes in ((1, [5]), (6, [0, 3, 0, 0, 4, 0]), (4, [2, 2, 2, 2]), (3, [0, 0, 0])):
        batch = torch.arange(n_mol).repeat_interleave(torch.tensor(sizes))
        ref = torch.zeros(n_mol + 1, dtype=torch.int64)
        ref[1:] = torch.tensor(sizes).cumsum(0)
        assert torch.equal(molecule_ptr(batch.cuda(), n_mol).cpu().long(), ref)


@pytest.mark.gpu
def test_in_flight_lanes_match_single_lane():
    """pipeline.InFlight: calls fed round-robin to three EnergyForces lanes on three HIP streams return the bits of the plain
    object, for interleaved DIFFERENT batches, after wait() -- starting from a COLD model (nothing packed yet)."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces, InFlight
    cfg, sd, head_sd, t = load_case("l2_sep_f32")
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    z, batch = t["z"].cuda(), t["batch"].cuda()
    g = torch.Generator().manual_seed(1)
    cases = []
    for _ in range(5):
        pos = (t["pos"] + 0.05 * torch.randn(t["pos"].shape, generator=g)).cuda()
        cases.append(distance(pos, batch, cfg["cutoff"], 32))
    # COLD lanes first: the lazily built operands (weight planes, backward transposes) are made by the first call of each kind
    # on one lane's stream while the other lanes are fenced off
    fl = InFlight(net, head, lanes=3, check_edges=False)
    out = [fl(z, ei, ed, ev, batch, cfg["n_mol"]) for ei, ed, ev in cases]
    out_e = [fl(z, ei, ed, ev, batch, cfg["n_mol"], forces=False) for ei, ed, ev in cases]
    fl.wait()
    torch.cuda.synchronize()
    plain = EnergyForces(net, head, check_edges=False)
    ref = [plain(z, ei, ed, ev, batch, cfg["n_mol"]) for ei, ed, ev in cases]
    for (e0, f0), (e1, f1), (e2, _) in zip(ref, out, out_e):
        assert torch.equal(e0, e1) and torch.equal(f0, f1) and torch.equal(e0, e2)


@pytest.mark.gpu
def test_in_flight_inputs_may_be_freed_right_after_the_call():
    """ADVICE r5 / VERDICT r5 item 5a: InFlight marks its tensor arguments as used on the lane's stream (record_stream) and the
    returned tensors as used on the waiting stream, so a data-loader loop that drops each batch right after the call -- and whose
    next batch lands in the recycled blocks -- still gets the bits of the keep-everything-alive run.  The caching allocator is
    ON (the default); each iteration allocates fresh inputs of the same sizes, i.e. exactly the blocks just freed."""
    from tests.test_hip_forces import _head_from_case
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces, InFlight
    cfg, sd, head_sd, t = load_case("l2_sep_f32")
    net, head = _mirror(cfg, sd), _head_from_case(cfg, head_sd)
    g = torch.Generator().manual_seed(3)
    pos_all = [t["pos"] + 0.05 * torch.randn(t["pos"].shape, generator=g) for _ in range(12)]
    plain = EnergyForces(net, head, check_edges=False, cache_topology=False)
    ref = []
    for pos in pos_all:
        z, batch = t["z"].cuda(), t["batch"].cuda()
        ref.append(plain(z, *distance(pos.cuda(), batch, cfg["cutoff"], 32), batch, cfg["n_mol"]))
    torch.cuda.synchronize()
    fl = InFlight(net, head, lanes=3, check_edges=False, cache_topology=False)
    out = []
    junk = None
    for pos in pos_all:
        z, batch = t["z"].cuda(), t["batch"].cuda()                 # fresh tensors every iteration ...
        ei, ed, ev = distance(pos.cuda(), batch, cfg["cutoff"], 32)
        out.append(fl(z, ei, ed, ev, batch, cfg["n_mol"]))
        del z, batch, ei, ed, ev                                      # ... dropped at once: their blocks are up for reuse
        junk = torch.full((1 << 16,), float("nan"), device="cuda")    # and something scribbles over recycled memory
        del junk
    fl.wait()
    keep = [(e.clone(), f.clone()) for e, f in out]                   # consumed on the current stream after wait()
    del out
    torch.cuda.synchronize()
    for (e0, f0), (e1, f1) in zip(ref, keep):
        assert torch.equal(e0, e1) and torch.equal(f0, f1)
    # a weight update between calls: the stale pack is rebuilt only after every lane has drained
    with torch.no_grad():
        net.gata_list[0].W_q.weight.mul_(1.0001)
    z, batch = t["z"].cuda(), t["batch"].cuda()
    ei, ed, ev = distance(pos_all[0].cuda(), batch, cfg["cutoff"], 32)
    e_new, f_new = fl(z, ei, ed, ev, batch, cfg["n_mol"])
    fl.wait()
    e_ref, f_ref = plain(z, ei, ed, ev, batch, cfg["n_mol"])
    assert torch.equal(e_new, e_ref) and torch.equal(f_new, f_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("agg,atomref", [("sum", True), ("mean", False), (None, False)])
def test_atomwise_n_out_3_matches_oracle(agg, atomref):
    """VERDICT r5 breadth: ``Atomwise(n_out > 1)`` (reference outputs.py:241, 323-376).  Energies [n_mol, 3], forces of the SUM of the
    outputs (``grad_outputs=ones``, outputs.py:365-375) through the fused pipeline and the reference-style call; and a NON-uniform
    upstream gradient through the autograd Function against the oracle's autograd."""
    import types
    import gotennet_amd
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    from oracle import gotennet_oracle as orc
    cfg, sd, _, t = load_case("l2_sep_f32")
    F, n_out = cfg["n_atom_basis"], 3
    torch.manual_seed(11)
    aref = torch.randn(cfg["max_z"], n_out) * 0.2 if atomref else None
    head = Atomwise(n_in=F, n_out=n_out, n_layers=2, n_hidden=24, aggregation_mode=agg, activation="silu", property="y",
                    contributions="yi", derivative="forces", mean=torch.tensor([0.3, -0.2, 0.1]), stddev=torch.tensor([1.7, 0.6, 1.1]),
                    atomref=aref)
    with torch.no_grad():
        for p in head.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    d = lambda m: {k: v.double() for k, v in m.items()}
    e_ref, f_ref, (h_ref, _, _) = orc.energy_and_forces(d(sd), cfg, d(hsd), t["z"], t["pos"].double(), t["batch"], cfg["n_mol"],
                                                       aggregation="mean" if agg == "mean" else "sum")
    yi_ref = orc.atomwise_contributions(d(hsd), h_ref, t["z"])
    net = gotennet_amd.GotenNetWrapper(
        n_atom_basis=F, n_interactions=cfg["n_interactions"], n_rbf=cfg["n_rbf"], cutoff_fn=gotennet_amd.CosineCutoff(cfg["cutoff"]),
        max_z=cfg["max_z"], num_heads=cfg["num_heads"], scale_edge=cfg["scale_edge"], lmax=cfg["lmax"],
        sep_dir=cfg["sep_dir"], sep_tensor=cfg["sep_tensor"])
    net.load_state_dict(sd, strict=True)
    net, head = net.cuda().eval(), head.cuda().eval()
    if agg is not None:
        e, f = EnergyForces(net, head)(t["z"].cuda(), t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda(),
                                       t["batch"].cuda(), cfg["n_mol"])
        assert e.shape == (cfg["n_mol"], n_out)
        assert rel_err(e.cpu(), e_ref) < TOL and rel_err(f.cpu(), f_ref) < TOL
    pos = t["pos"].cuda().requires_grad_(True)
    inp = types.SimpleNamespace(z=t["z"].cuda(), pos=pos, batch=t["batch"].cuda())
    inp.representation, inp.vector_representation = net(inp)
    out = head(inp)
    assert rel_err(out["yi"].cpu(), yi_ref) < TOL
    if agg is None:
        assert out["y"].shape == (t["z"].shape[0], n_out) and rel_err(out["y"].cpu(), yi_ref) < TOL
        # per-atom outputs: the oracle's forces of sum(y_i) (its "sum" aggregation differentiates the same scalar)
        assert rel_err(out["forces"].cpu(), f_ref) < TOL
    else:
        assert rel_err(out["y"].cpu(), e_ref) < TOL and rel_err(out["forces"].cpu(), f_ref) < TOL
        # a non-uniform upstream gradient on the [n_mol, 3] property
        wgt = torch.tensor([[1.0, -2.0, 0.5]]) * (1.0 + torch.arange(cfg["n_mol"]).float().unsqueeze(1))
        pos2 = t["pos"].cuda().requires_grad_(True)
        inp2 = types.SimpleNamespace(z=t["z"].cuda(), pos=pos2, batch=t["batch"].cuda())
        inp2.representation, inp2.vector_representation = net(inp2)
        head.derivative = None
        (gp,) = torch.autograd.grad((head(inp2)["y"] * wgt.cuda()).sum(), pos2)
        pr = t["pos"].double().clone().requires_grad_(True)
        ei, w, vec = orc.distance(pr, t["batch"], cfg["cutoff"], 32)
        hh, _ = orc.gotennet_forward(d(sd), cfg, t["z"], ei, w, vec)
        er = orc.atomwise_energy(d(hsd), hh, t["batch"], cfg["n_mol"], z=t["z"], aggregation="mean" if agg == "mean" else "sum")
        (gr,) = torch.autograd.grad((er * wgt.double()).sum(), pr)
        assert rel_err(gp.cpu(), gr) < TOL

