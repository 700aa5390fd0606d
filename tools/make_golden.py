#!/usr/bin/env python
"""Generate golden input/output vectors from the REAL reference.

Runs only in the survey/build container (needs /root/reference and
tools/ref_shims.py).  Writes small ``.npz`` fixtures under ``tests/golden/``:
inputs, the full reference ``state_dict`` (randomised, including biases and
LayerNorm affine so nothing is hidden by zero-initialisation), the reference's
per-layer and final outputs in fp32, the same forward evaluated by the
reference in fp64 (the "truth" used to rank fp32 implementations), and energy
/ forces through the reference ``Atomwise`` head with autograd.

    python tools/make_golden.py            # regenerate every fixture
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref = ref_shims.import_reference()
from gotennet.models.components import layers as ref_layers  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# name -> (hyper-parameters, graph spec)
CASES = {
    # repo-yaml flavour (sep_dir/sep_tensor, scale_edge off), l = 2
    "l2_sep_f32": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                        sep_dir=True, sep_tensor=True, max_z=10), dict(mols=[7, 5], box=3.2, seed=1)),
    # class-default flavour (no sep, scale_edge on), l = 1
    "l1_nosep_scale_f32": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=1, num_heads=4, scale_edge=True,
                                sep_dir=False, sep_tensor=False, max_z=10), dict(mols=[6, 6], box=3.0, seed=2)),
    "l3_sep_scale_f32": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=3, num_heads=8, scale_edge=True,
                              sep_dir=True, sep_tensor=True, max_z=10), dict(mols=[5, 4], box=2.8, seed=3)),
    "l4_sep_f32": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=4, num_heads=8, scale_edge=False,
                        sep_dir=True, sep_tensor=True, max_z=10), dict(mols=[6], box=2.6, seed=4)),
    # mixed flags + wider features + a molecule partly outside the cutoff + isolated atom
    "l2_mixed_f64ch": (dict(n_atom_basis=64, n_interactions=2, n_rbf=16, lmax=2, num_heads=8, scale_edge=True,
                            sep_dir=True, sep_tensor=False, max_z=10), dict(mols=[9, 1, 4], box=6.5, seed=5)),
    # edge list shuffled (not target-sorted) and without self-loops
    "l2_sep_shuffled_noloop": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=2, num_heads=8,
                                    scale_edge=False, sep_dir=True, sep_tensor=True, max_z=10),
                               dict(mols=[6, 5], box=3.0, seed=6, shuffle=True, loop=False)),
    # ---- non-default flags --------------------------------------------------------
    "opt_bessel_norej": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                              sep_dir=True, sep_tensor=True, max_z=10, radial_basis="BesselBasis",
                              edge_updates="norej"), dict(mols=[6, 5], box=3.0, seed=11)),
    "opt_gauss_jointhtr_gated": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=2, num_heads=8,
                                      scale_edge=True, sep_dir=True, sep_tensor=True, max_z=10,
                                      radial_basis="GaussianRBF", sep_htr=False, edge_updates="gated"),
                                 dict(mols=[6, 4], box=3.0, seed=12)),
    "opt_jointhtr_l3_tanh": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=3, num_heads=8, scale_edge=False,
                                  sep_dir=True, sep_tensor=True, max_z=10, sep_htr=False, edge_updates="gatedt_norm"),
                             dict(mols=[5, 4], box=2.8, seed=13)),
    "opt_noupd_act": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=1, num_heads=4, scale_edge=True,
                           sep_dir=False, sep_tensor=False, max_z=10, edge_updates=False),
                      dict(mols=[6, 5], box=3.0, seed=14)),
    "opt_act_norej_joint": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                                 sep_dir=False, sep_tensor=True, max_z=10, sep_htr=False, edge_updates="act_norej"),
                            dict(mols=[7, 3], box=3.0, seed=15)),
    "opt_layernorm_tln": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                               sep_dir=True, sep_tensor=True, max_z=10, layernorm="layer", steerable_norm="tensor"),
                          dict(mols=[6, 5], box=3.0, seed=16)),
    "opt_tln_l4": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=4, num_heads=8, scale_edge=True,
                        sep_dir=True, sep_tensor=True, max_z=10, steerable_norm="tensor"),
                   dict(mols=[5], box=2.6, seed=17)),
    # composed edge updates; activation passed as a string so that "linwa" can put it into nn.Sequential
    "opt_mlp_linwa_ln_gated": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                                    sep_dir=True, sep_tensor=True, max_z=10, activation="silu",
                                    edge_updates="mlp_linwa_ln_gated", edge_ln="layer"),
                               dict(mols=[6, 5], box=3.0, seed=21)),
    "opt_mlpa_linw_postln": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=1, num_heads=4, scale_edge=True,
                                  sep_dir=False, sep_tensor=False, max_z=10, activation="silu",
                                  edge_updates="mlpa_linw_postln_norej", sep_htr=False),
                             dict(mols=[7, 4], box=3.0, seed=22)),
    "opt_linw_act_l3": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=3, num_heads=8, scale_edge=False,
                             sep_dir=True, sep_tensor=True, max_z=10, activation="silu", edge_updates="linw_act"),
                        dict(mols=[5, 4], box=2.8, seed=23)),
    # non-SiLU activations (str2act names, layers.py:596-700); "softplus" is the reference's shifted softplus
    "opt_act_ssp": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                         sep_dir=True, sep_tensor=True, max_z=10, activation="softplus"),
                    dict(mols=[6, 5], box=3.0, seed=31)),
    "opt_act_tanh_l3_gated": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=3, num_heads=8, scale_edge=True,
                                   sep_dir=True, sep_tensor=True, max_z=10, activation="tanh", edge_updates="gated",
                                   layernorm="layer"), dict(mols=[5, 4], box=2.8, seed=32)),
    "opt_act_gelu_mlpa_linwa": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                                     sep_dir=True, sep_tensor=True, max_z=10, activation="gelu",
                                     edge_updates="mlpa_linwa", edge_ln="layer"), dict(mols=[6, 4], box=3.0, seed=33)),
    "opt_act_mish_l4": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=4, num_heads=8, scale_edge=False,
                             sep_dir=True, sep_tensor=True, max_z=10, activation="mish"),
                        dict(mols=[6], box=2.6, seed=34)),
    # degrees 5..8 (TensorInit's recursion beyond l = 4, layers.py:934-1494; every per-degree structure at its widest)
    "l5_sep_f32": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=5, num_heads=8, scale_edge=False,
                        sep_dir=True, sep_tensor=True, max_z=10), dict(mols=[5], box=2.6, seed=41)),
    "l6_nosep_scale_f32": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=6, num_heads=4, scale_edge=True,
                                sep_dir=False, sep_tensor=False, max_z=10), dict(mols=[4, 3], box=2.6, seed=42)),
    "l7_mixed_jointhtr_gated": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=7, num_heads=8, scale_edge=False,
                                     sep_dir=True, sep_tensor=False, max_z=10, sep_htr=False, edge_updates="gated"),
                                dict(mols=[5], box=2.6, seed=43)),
    "l8_sep_tln_f32": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=8, num_heads=8, scale_edge=True,
                            sep_dir=True, sep_tensor=True, max_z=10, steerable_norm="tensor"),
                       dict(mols=[4], box=2.4, seed=44)),
    # aggr != "add" (gotennet.py:84,638: the PyG reduce of GATA.aggregate); "max" has no HIP backward (forward-only fixture use)
    "opt_aggr_mean_l2": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                              sep_dir=True, sep_tensor=True, max_z=10, aggr="mean"), dict(mols=[6, 1, 5], box=3.0, seed=51)),
    "opt_aggr_mean_l5_nosep": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=5, num_heads=4, scale_edge=True,
                                    sep_dir=False, sep_tensor=False, max_z=10, aggr="mean"), dict(mols=[5, 3], box=2.6, seed=52)),
    "opt_aggr_max_l3": (dict(n_atom_basis=32, n_interactions=2, n_rbf=8, lmax=3, num_heads=8, scale_edge=False,
                             sep_dir=True, sep_tensor=True, max_z=10, aggr="max"), dict(mols=[6, 4], box=2.8, seed=53)),
    "opt_evec16_emlp48": (dict(n_atom_basis=32, n_interactions=3, n_rbf=8, lmax=2, num_heads=8, scale_edge=False,
                               sep_dir=True, sep_tensor=True, max_z=10, activation="silu",
                               edge_updates="mlpa_linwa_postln_gatedt", edge_ln="layer", evec_dim=16, emlp_dim=48),
                          dict(mols=[6, 5], box=3.0, seed=24)),
}

# full-width models (BASELINE configs[0] and the configs[1] model): weights come from tests.golden_util.seeded_fill,
# the fixture stores inputs and reference outputs only
SEEDED = {
    # C1: QM9_small -- F=128, L=4, lmax=2, one 19-atom molecule, yaml flags
    "c1_qm9_small_seeded": (dict(n_atom_basis=128, n_interactions=4, n_rbf=32, lmax=2, num_heads=8, scale_edge=False,
                                 sep_dir=True, sep_tensor=True, max_z=10), dict(mols=[19], box=4.0, seed=31), 71, 64),
    # C2 model (F=256, L=6, lmax=2) on three 21-atom molecules
    "c2_model_3mol_seeded": (dict(n_atom_basis=256, n_interactions=6, n_rbf=32, lmax=2, num_heads=8, scale_edge=False,
                                  sep_dir=True, sep_tensor=True, max_z=10), dict(mols=[21, 21, 21], box=4.6, seed=32), 72, 256),
    # the north-star's "L=4" reading: the same model at lmax=4, one 21-atom molecule
    "c2_model_lmax4_1mol_seeded": (dict(n_atom_basis=256, n_interactions=6, n_rbf=32, lmax=4, num_heads=8, scale_edge=False,
                                        sep_dir=True, sep_tensor=True, max_z=10), dict(mols=[21], box=4.6, seed=33), 73, 256),
}

CUTOFF = 5.0


def make_molecules(spec):
    g = torch.Generator().manual_seed(spec["seed"])
    pos, batch, z = [], [], []
    for b, n in enumerate(spec["mols"]):
        pos.append(torch.rand((n, 3), generator=g) * spec["box"] + 10.0 * b)
        batch += [b] * n
        z.append(torch.randint(1, 9, (n,), generator=g))
    return torch.cat(pos), torch.tensor(batch), torch.cat(z)


def randomise(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("norm.weight") and "tensor_layernorm" not in name:
                p.copy_(1.0 + 0.2 * (torch.rand(p.shape, generator=g) - 0.5))
            elif p.dim() == 1:  # biases
                p.copy_(0.1 * (torch.rand(p.shape, generator=g) - 0.5))
            elif "A_na" in name or "A_nbr" in name:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
                if "A_na" in name:
                    p[0].zero_()  # padding_idx row
            else:
                fan_out, fan_in = p.shape
                a = (6.0 / (fan_in + fan_out)) ** 0.5  # xavier-uniform bound
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * a)
        for name, b in module.named_buffers():
            if name.endswith("tensor_layernorm.weight"):   # a buffer of ones in the reference; loaded from checkpoints
                b.copy_(1.0 + 0.2 * (torch.rand(b.shape, generator=g) - 0.5))


def run_reference(net, z, ei, w, vec, trace=False):
    """Reference forward; returns h, X and (optionally) per-layer (h, X, t)."""
    layers = []
    hooks = []
    if trace:
        state = {}

        def gata_hook(_m, _i, out):
            state["t"] = out[2]

        def eqff_hook(_m, _i, out):
            layers.append((out[0].squeeze(1).detach().clone(), out[1].detach().clone(), state["t"].detach().clone()))

        for g_, e_ in zip(net.gata_list, net.eqff_list):
            hooks.append(g_.register_forward_hook(gata_hook))
            hooks.append(e_.register_forward_hook(eqff_hook))
    h, X = net(z, ei, w.clone(), vec.clone())  # clone: the reference normalises edge_vec in place
    for hk in hooks:
        hk.remove()
    return h, X, layers


def build(name, hp, spec):
    torch.manual_seed(0)
    net = ref.GotenNet(cutoff_fn=ref_layers.CosineCutoff(CUTOFF), **hp)
    randomise(net, 1000 + spec["seed"])
    net.eval()
    pos, batch, z = make_molecules(spec)
    n_mol = len(spec["mols"])

    dist = ref_layers.Distance(CUTOFF, max_num_neighbors=32, loop=spec.get("loop", True))
    ei, w, vec = dist(pos, batch)
    if spec.get("shuffle"):
        perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(99))
        ei, w, vec = ei[:, perm], w[perm], vec[perm]

    with torch.no_grad():
        h, X, layers = run_reference(net, z, ei, w, vec, trace=True)
        phi = net.radial_basis(w)
        mask = ei[0] != ei[1]
        unit = vec.clone()
        unit[mask] = unit[mask] / torch.norm(unit[mask], dim=1, keepdim=True)
        rl = net.sphere(unit)
        net64 = ref.GotenNet(cutoff_fn=ref_layers.CosineCutoff(CUTOFF), **hp).double()
        net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
        net64.eval()
        h64, X64, _ = run_reference(net64, z, ei, w.double(), vec.double())

    # energy + forces through the reference Atomwise head (outputs.py:323-376); only
    # for loop=True unshuffled cases (the wrapper path).
    extra = {}
    if spec.get("loop", True) and not spec.get("shuffle"):
        sys.modules.setdefault("torch_scatter", types.ModuleType("torch_scatter"))
        sys.modules["torch_scatter"].scatter = ref_shims._scatter
        sys.modules.setdefault("ase", types.ModuleType("ase"))
        sys.modules.setdefault("ase.data", types.ModuleType("ase.data"))
        sys.modules["ase.data"].atomic_masses = np.ones(120)
        sys.modules["ase"].data = sys.modules["ase.data"]
        from gotennet.models.components import outputs as ref_out
        head = ref_out.Atomwise(n_in=hp["n_atom_basis"], n_hidden=16, activation=torch.nn.functional.silu,
                                property="property", derivative="forces")
        randomise(head, 2000 + spec["seed"])
        for tag, nn_, hd_, dt in (("", net, head, torch.float32),
                                  ("_f64", net64, ref_out.Atomwise(n_in=hp["n_atom_basis"], n_hidden=16,
                                                                   activation=torch.nn.functional.silu,
                                                                   property="property", derivative="forces").double(), torch.float64)):
            if tag:
                hd_.load_state_dict({k: v.double() for k, v in head.state_dict().items()})
            p = pos.to(dt).clone().requires_grad_(True)
            if dt == torch.float32:
                dist_ = ref_layers.Distance(CUTOFF, max_num_neighbors=32, loop=True)
                ei_, w_, vec_ = dist_(p, batch)
                assert torch.equal(ei_, ei)
            else:  # Distance.forward (layers.py:1588-1604) allocates fp32; same arithmetic in fp64
                ei_ = ei
                vec_ = p[ei_[0]] - p[ei_[1]]
                m_ = ei_[0] != ei_[1]
                w_ = torch.zeros(vec_.size(0), dtype=dt)
                w_[m_] = torch.norm(vec_[m_], dim=-1)
            hh, XX = nn_(z, ei_, w_, vec_ * 1.0)  # fresh non-leaf: the reference writes edge_vec in place

            class _D(dict):
                __getattr__ = dict.__getitem__
            inp = _D(z=z, pos=p, batch=batch, representation=hh, vector_representation=XX)
            res = hd_(inp)
            extra["energy" + tag] = res["property"].detach().numpy()
            extra["forces" + tag] = res["forces"].detach().numpy()
        for k, v in head.state_dict().items():
            extra["head/" + k] = v.numpy()

    arrays = dict(
        z=z.numpy(), pos=pos.numpy(), batch=batch.numpy(), edge_index=ei.numpy(),
        edge_diff=w.numpy(), edge_vec=vec.numpy(),
        h=h.numpy(), X=X.numpy(), h_f64=h64.numpy(), X_f64=X64.numpy(),
        phi=phi.numpy(), rl=rl.numpy(),
        cfg=np.frombuffer(json.dumps({**dict(cutoff=CUTOFF, epsilon=1e-8, sep_htr=True, n_mol=n_mol), **hp}).encode(), dtype=np.uint8),
    )
    for li, (lh, lX, lt) in enumerate(layers):
        arrays[f"layer{li}/h"] = lh.numpy()
        arrays[f"layer{li}/X"] = lX.numpy()
        arrays[f"layer{li}/t"] = lt.numpy()
    for k, v in net.state_dict().items():
        arrays["sd/" + k] = v.numpy()
    arrays.update(extra)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: N={len(z)} E={ei.shape[1]} |h|max={h.abs().max():.3f} |X|max={X.abs().max():.4f} "
          f"fp32-vs-fp64 dh={float((h.double()-h64).abs().max()):.2e} -> {os.path.getsize(path)/1024:.0f} KiB")


def build_seeded(name, hp, spec, wseed, head_hidden):
    """Reference outputs (fp32 and fp64, energy/forces through the reference Atomwise) for seeded weights."""
    sys.path.insert(0, ROOT)
    from tests.golden_util import seeded_fill
    sys.modules.setdefault("torch_scatter", types.ModuleType("torch_scatter"))
    sys.modules["torch_scatter"].scatter = ref_shims._scatter
    sys.modules.setdefault("ase", types.ModuleType("ase"))
    sys.modules.setdefault("ase.data", types.ModuleType("ase.data"))
    sys.modules["ase.data"].atomic_masses = np.ones(120)
    sys.modules["ase"].data = sys.modules["ase.data"]
    from gotennet.models.components import outputs as ref_out
    pos, batch, z = make_molecules(spec)
    out = {}
    for tag, dt in (("", torch.float32), ("_f64", torch.float64)):
        torch.manual_seed(0)
        net = ref.GotenNet(cutoff_fn=ref_layers.CosineCutoff(CUTOFF), **hp)
        head = ref_out.Atomwise(n_in=hp["n_atom_basis"], n_hidden=head_hidden, activation=torch.nn.functional.silu,
                                property="property", derivative="forces")
        seeded_fill(net, wseed)
        seeded_fill(head, wseed + 1)
        net, head = net.to(dt).eval(), head.to(dt).eval()
        p = pos.to(dt).clone().requires_grad_(True)
        ei, w32, vec32 = ref_layers.Distance(CUTOFF, max_num_neighbors=32, loop=True)(pos, batch)
        vec = p[ei[0]] - p[ei[1]]
        m_ = ei[0] != ei[1]
        w = torch.zeros(vec.size(0), dtype=dt)
        w[m_] = torch.norm(vec[m_], dim=-1)
        hh, XX = net(z, ei, w, vec * 1.0)

        class _D(dict):
            __getattr__ = dict.__getitem__
        res = head(_D(z=z, pos=p, batch=batch, representation=hh, vector_representation=XX))
        with torch.no_grad():                         # (h, X) from the fp32 edge inputs cast to the run's dtype
            h_in, X_in = net(z, ei, w32.to(dt).clone(), vec32.to(dt).clone())
        out["h" + tag], out["X" + tag] = h_in.numpy(), X_in.numpy()
        out["energy" + tag], out["forces" + tag] = res["property"].detach().numpy(), res["forces"].detach().numpy()
        if not tag:
            out.update(edge_index=ei.numpy(), edge_diff=w32.numpy(), edge_vec=vec32.numpy())
    cfg = {**dict(cutoff=CUTOFF, epsilon=1e-8, sep_htr=True, n_mol=len(spec["mols"]), seeded=wseed, head_hidden=head_hidden), **hp}
    arrays = dict(z=z.numpy(), pos=pos.numpy(), batch=batch.numpy(),
                  cfg=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), **out)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: N={len(z)} E={out['edge_index'].shape[1]} |h|max={np.abs(out['h']).max():.3f} "
          f"fp32-vs-fp64 dh={np.abs(out['h'] - out['h_f64']).max():.2e} -> {os.path.getsize(path)/1024:.0f} KiB")


def build_full_forward(name="c2_full_forward_seeded", wseed=72, n_mol=128):
    """BASELINE configs[1] at FULL size (128 aspirin-like molecules, the bench inputs, the C2 model with seeded weights):
    the reference's forward and per-molecule energies on the CPU (forces need > 62 GB of autograd state there).  The
    fixture keeps sampled rows and column sums of (h, X) and all energies; inputs are regenerated from their seeds."""
    sys.path.insert(0, ROOT)
    from tests.golden_util import seeded_fill
    from gotennet_amd import synthetic
    sys.modules.setdefault("torch_scatter", types.ModuleType("torch_scatter"))
    sys.modules["torch_scatter"].scatter = ref_shims._scatter
    sys.modules.setdefault("ase", types.ModuleType("ase"))
    sys.modules.setdefault("ase.data", types.ModuleType("ase.data"))
    sys.modules["ase.data"].atomic_masses = np.ones(120)
    sys.modules["ase"].data = sys.modules["ase.data"]
    from gotennet.models.components import outputs as ref_out
    hp = dict(n_atom_basis=256, n_interactions=6, n_rbf=32, lmax=2, num_heads=8, scale_edge=False, sep_dir=True,
              sep_tensor=True, max_z=10)
    pos, batch, z = synthetic.make_batch("rmd17_aspirin", n_mol, seed=0)
    torch.manual_seed(0)
    net = ref.GotenNet(cutoff_fn=ref_layers.CosineCutoff(CUTOFF), **hp)
    head = ref_out.Atomwise(n_in=256, n_hidden=256, activation=torch.nn.functional.silu, property="property")
    seeded_fill(net, wseed)
    seeded_fill(head, wseed + 1)
    net, head = net.eval(), head.eval()
    torch.set_num_threads(8)
    with torch.no_grad():
        ei, w, vec = ref_layers.Distance(CUTOFF, max_num_neighbors=32, loop=True)(pos, batch)
        h, X = net(z, ei, w.clone(), vec.clone())

        class _D(dict):
            __getattr__ = dict.__getitem__
        e = head(_D(z=z, pos=pos, batch=batch, representation=h, vector_representation=X))["property"]
    # energy + forces by the reference's own autograd for the first n_force molecules (a quarter of the batch fits
    # in this container's 62 GB; molecules are independent, so they are the full batch's values for those atoms)
    n_force = 32
    na = len(z) // n_mol
    pf = pos[: n_force * na].clone().requires_grad_(True)
    bf, zf_ = batch[: n_force * na], z[: n_force * na]
    head_f = ref_out.Atomwise(n_in=256, n_hidden=256, activation=torch.nn.functional.silu, property="property",
                              derivative="forces")
    head_f.load_state_dict(head.state_dict())
    ei_f, _, _ = ref_layers.Distance(CUTOFF, max_num_neighbors=32, loop=True)(pf.detach(), bf)
    vec_f = pf[ei_f[0]] - pf[ei_f[1]]
    m_f = ei_f[0] != ei_f[1]
    w_f = torch.zeros(vec_f.size(0))
    w_f[m_f] = torch.norm(vec_f[m_f], dim=-1)
    hf, Xf = net(zf_, ei_f, w_f, vec_f * 1.0)
    res_f = head_f(_D(z=zf_, pos=pf, batch=bf, representation=hf, vector_representation=Xf))
    forces_part = res_f["forces"].detach()
    assert torch.allclose(res_f["property"].detach(), e[:n_force], rtol=1e-5, atol=1e-5)
    rows_h, rows_X = torch.arange(0, len(z), 37), torch.arange(0, len(z), 149)
    cfg = {**dict(cutoff=CUTOFF, epsilon=1e-8, sep_htr=True, n_mol=n_mol, seeded=wseed, head_hidden=256,
                  workload="rmd17_aspirin", batch_seed=0), **hp}
    arrays = dict(cfg=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), n_edges=np.array(ei.shape[1]),
                  edge_index_checksum=np.array([int(ei[0].sum()), int(ei[1].sum()), int((ei[0] * 31 + ei[1]).sum() % (2 ** 61))]),
                  rows_h=rows_h.numpy(), h_rows=h[rows_h].numpy(), rows_X=rows_X.numpy(), X_rows=X[rows_X].numpy(),
                  h_colsum=h.double().sum(0).numpy(), X_colsum=X.double().sum(0).numpy(),
                  h_abs_sum=np.array(float(h.double().abs().sum())), X_abs_sum=np.array(float(X.double().abs().sum())),
                  energy=e.numpy(), n_force_molecules=np.array(n_force), forces_part=forces_part.numpy())
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: N={len(z)} E={ei.shape[1]} |h|max={float(h.abs().max()):.3f} -> {os.path.getsize(path)/1024:.0f} KiB")


# BASELINE configs[2] / configs[4] workloads (gotennet_amd.synthetic, the bench inputs) through the reference with the
# seeded full-width models: name -> (hyper-parameters, workload, molecules, weight seed, head hidden, row stride h / X)
WORKLOAD_CASES = {
    # C3: MD22 Ac-Ala3-NHMe-like (42 atoms), the C2 model, energy + forces; molecules 0..1 of the bench batch
    "c3_ac_ala3_2mol_seeded": (dict(n_atom_basis=256, n_interactions=6, n_rbf=32, lmax=2, num_heads=8, scale_edge=False,
                                    sep_dir=True, sep_tensor=True, max_z=10), "md22_ac_ala3", 2, 74, 256, 1, 3),
    # C5: MD22 double-walled-nanotube-like (370 atoms, ~63 neighbours inside 5 A -> the 32-neighbour cap of
    # Distance.forward (layers.py:1588-1604) is active on almost every atom), F=256, L=6, lmax=3; molecule 0 of the bench batch
    "c5_nanotube_1mol_seeded": (dict(n_atom_basis=256, n_interactions=6, n_rbf=32, lmax=3, num_heads=8, scale_edge=False,
                                     sep_dir=True, sep_tensor=True, max_z=10), "md22_nanotube", 1, 75, 256, 7, 23),
}


def build_workload(name, hp, workload, n_mol, wseed, head_hidden, stride_h, stride_X):
    """Reference (h, X), energy and forces on molecules of a gotennet_amd.synthetic workload (the bench inputs), radius
    graph by the reference's Distance (neighbour cap included).  Stores the full edge list, energies, forces, sampled
    rows + column sums of (h, X) in fp32, and the same rows from an fp64 forward."""
    sys.path.insert(0, ROOT)
    from tests.golden_util import seeded_fill
    from gotennet_amd import synthetic
    sys.modules.setdefault("torch_scatter", types.ModuleType("torch_scatter"))
    sys.modules["torch_scatter"].scatter = ref_shims._scatter
    sys.modules.setdefault("ase", types.ModuleType("ase"))
    sys.modules.setdefault("ase.data", types.ModuleType("ase.data"))
    sys.modules["ase.data"].atomic_masses = np.ones(120)
    sys.modules["ase"].data = sys.modules["ase.data"]
    from gotennet.models.components import outputs as ref_out
    torch.set_num_threads(8)
    pos, batch, z = synthetic.make_batch(workload, n_mol, seed=0)
    torch.manual_seed(0)
    net = ref.GotenNet(cutoff_fn=ref_layers.CosineCutoff(CUTOFF), **hp)
    head = ref_out.Atomwise(n_in=hp["n_atom_basis"], n_hidden=head_hidden, activation=torch.nn.functional.silu,
                            property="property", derivative="forces")
    seeded_fill(net, wseed)
    seeded_fill(head, wseed + 1)
    net, head = net.eval(), head.eval()
    dist = ref_layers.Distance(CUTOFF, max_num_neighbors=32, loop=True)
    ei, w32, vec32 = dist(pos, batch)
    deg = torch.bincount(ei[1], minlength=len(z))
    p = pos.clone().requires_grad_(True)
    vec = p[ei[0]] - p[ei[1]]
    m_ = ei[0] != ei[1]
    w = torch.zeros(vec.size(0))
    w[m_] = torch.norm(vec[m_], dim=-1)
    hh, XX = net(z, ei, w, vec * 1.0)

    class _D(dict):
        __getattr__ = dict.__getitem__
    res = head(_D(z=z, pos=p, batch=batch, representation=hh, vector_representation=XX))
    energy, forces = res["property"].detach(), res["forces"].detach()
    del res, hh, XX
    with torch.no_grad():
        h, X = net(z, ei, w32.clone(), vec32.clone())
        net64 = ref.GotenNet(cutoff_fn=ref_layers.CosineCutoff(CUTOFF), **hp)
        seeded_fill(net64, wseed)
        net64 = net64.double().eval()
        h64, X64 = net64(z, ei, w32.double(), vec32.double())
    rows_h, rows_X = torch.arange(0, len(z), stride_h), torch.arange(0, len(z), stride_X)
    cfg = {**dict(cutoff=CUTOFF, epsilon=1e-8, sep_htr=True, n_mol=n_mol, seeded=wseed, head_hidden=head_hidden,
                  workload=workload, batch_seed=0), **hp}
    arrays = dict(cfg=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
                  edge_index=ei.numpy().astype(np.int32), max_in_degree=np.array(int(deg.max())),
                  rows_h=rows_h.numpy(), h_rows=h[rows_h].numpy(), h_rows_f64=h64[rows_h].numpy(),
                  rows_X=rows_X.numpy(), X_rows=X[rows_X].numpy(), X_rows_f64=X64[rows_X].numpy(),
                  h_colsum=h.double().sum(0).numpy(), X_colsum=X.double().sum(0).numpy(),
                  h_abs_sum=np.array(float(h.double().abs().sum())), X_abs_sum=np.array(float(X.double().abs().sum())),
                  energy=energy.numpy(), forces=forces.numpy())
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: N={len(z)} E={ei.shape[1]} max in-degree {int(deg.max())} |h|max={float(h.abs().max()):.3f} "
          f"|X|max={float(X.abs().max()):.4f} fp32-vs-fp64 dh={float((h.double() - h64).abs().max()):.2e} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def sh_kat():
    """Known-answer table for TensorInit (layers.py:805-902) on fixed unit vectors, l <= 4,
    and for ExpNormalSmearing/CosineCutoff incl. d = 0 and d >= cutoff."""
    g = torch.Generator().manual_seed(7)
    v = torch.randn((32, 3), generator=g, dtype=torch.float64)
    v = v / v.norm(dim=1, keepdim=True)
    v = torch.cat([v, torch.eye(3, dtype=torch.float64), torch.zeros((1, 3), dtype=torch.float64)])
    out = {"unit": v.numpy()}
    for l in (1, 2, 3, 4):
        out[f"sh{l}"] = ref_layers.TensorInit(l=l)(v).numpy()
    d = torch.tensor([0.0, 0.3, 1.0, 2.5, 4.0, 4.999, 5.0, 5.5], dtype=torch.float64)
    for R in (8, 32):
        rb = ref_layers.ExpNormalSmearing(cutoff=CUTOFF, n_rbf=R)
        out[f"rbf{R}_means"] = rb.means.numpy()
        out[f"rbf{R}_betas"] = rb.betas.numpy()
        out[f"rbf{R}"] = rb.double()(d).numpy()
    out["d"] = d.numpy()
    out["cut"] = ref_layers.CosineCutoff(CUTOFF)(d).numpy()
    np.savez_compressed(os.path.join(OUT, "kat_basis.npz"), **out)
    print("kat_basis written")


def sh_kat_high():
    """Known-answer table for TensorInit at l = 5..8 (layers.py:934-1494) on unit vectors, the axes, a zero vector
    and a few NON-unit vectors (the formulas are polynomials; the recursion must match off the sphere too)."""
    g = torch.Generator().manual_seed(8)
    v = torch.randn((40, 3), generator=g, dtype=torch.float64)
    v[:32] = v[:32] / v[:32].norm(dim=1, keepdim=True)
    v = torch.cat([v, torch.eye(3, dtype=torch.float64), torch.zeros((1, 3), dtype=torch.float64)])
    out = {"vec": v.numpy()}
    for l in (5, 6, 7, 8):
        out[f"sh{l}"] = ref_layers.TensorInit(l=l)(v).numpy()
    np.savez_compressed(os.path.join(OUT, "kat_sh_l8.npz"), **out)
    print("kat_sh_l8 written")


def head_kat():
    """Known-answer test for the reference Atomwise head (outputs.py:323-376) with NON-trivial mean / stddev / atomref:
    random atom features in, per-molecule property and per-atom contributions out."""
    sys.modules.setdefault("torch_scatter", types.ModuleType("torch_scatter"))
    sys.modules["torch_scatter"].scatter = ref_shims._scatter
    sys.modules.setdefault("ase", types.ModuleType("ase"))
    sys.modules.setdefault("ase.data", types.ModuleType("ase.data"))
    sys.modules["ase.data"].atomic_masses = np.ones(120)
    sys.modules["ase"].data = sys.modules["ase.data"]
    from gotennet.models.components import outputs as ref_out
    g = torch.Generator().manual_seed(41)
    F_, Hd, n_mol = 64, 32, 3
    sizes = [9, 1, 14]
    z = torch.randint(1, 10, (sum(sizes),), generator=g)
    batch = torch.repeat_interleave(torch.arange(n_mol), torch.tensor(sizes))
    h = torch.randn((len(z), F_), generator=g)
    atomref = torch.randn((10, 1), generator=g) * 3.0
    head = ref_out.Atomwise(n_in=F_, n_hidden=Hd, activation=torch.nn.functional.silu, property="property",
                            contributions="contrib", mean=torch.tensor([1.7]), stddev=torch.tensor([0.35]),
                            atomref=atomref)
    randomise(head, 4100)

    class _D(dict):
        __getattr__ = dict.__getitem__
    with torch.no_grad():
        res = head(_D(z=z, batch=batch, representation=h, vector_representation=None))
    out = dict(h=h.numpy(), z=z.numpy(), batch=batch.numpy(), n_mol=np.array(n_mol), atomref=atomref.numpy(),
               energy=res["property"].numpy(), contrib=res["contrib"].numpy())
    for k, v in head.state_dict().items():
        out["head/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "kat_head.npz"), **out)
    print("kat_head written:", sorted(k for k in out if k.startswith("head/")))
    # a deeper head with the reference's DEFAULT activation (shifted softplus), a pyramid of hidden widths and "mean"
    head3 = ref_out.Atomwise(n_in=F_, n_layers=3, aggregation_mode="mean", property="property", contributions="contrib",
                             mean=torch.tensor([-0.4]), stddev=torch.tensor([2.5]))
    randomise(head3, 4200)
    with torch.no_grad():
        res3 = head3(_D(z=z, batch=batch, representation=h, vector_representation=None))
    out3 = dict(h=h.numpy(), z=z.numpy(), batch=batch.numpy(), n_mol=np.array(n_mol), energy=res3["property"].numpy(),
                contrib=res3["contrib"].numpy())
    for k, v in head3.state_dict().items():
        out3["head/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "kat_head_l3_mean_ssp.npz"), **out3)
    print("kat_head_l3_mean_ssp written:", [tuple(v.shape) for k, v in head3.state_dict().items() if k.endswith("weight")])
    # AtomwiseV3 (outputs.py:96-229): scale per atom, mean added AFTER the aggregation; "sum", "mean" and None
    outv = dict(h=h.numpy(), z=z.numpy(), batch=batch.numpy(), n_mol=np.array(n_mol), atomref=atomref.numpy(),
                mean=np.array(1.7), stddev=np.array(0.35))
    for tag, agg in (("sum", "sum"), ("mean", "mean"), ("none", None)):
        hv = ref_out.AtomwiseV3(n_in=F_, n_hidden=Hd, activation=torch.nn.functional.silu, property="property",
                                contributions="contrib", mean=1.7, stddev=0.35, atomref=atomref, aggregation_mode=agg)
        randomise(hv, 4300)
        with torch.no_grad():
            resv = hv(_D(z=z, batch=batch, representation=h, vector_representation=None))
        outv[f"energy_{tag}"] = resv["property"].numpy()
        outv[f"contrib_{tag}"] = resv["contrib"].numpy()
        for k, v in hv.state_dict().items():
            outv["head/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "kat_head_v3.npz"), **outv)
    print("kat_head_v3 written:", sorted(k for k in outv if not k.startswith("head/")))


def qm9_heads_kat():
    """Known-answer tests for the two vector read-outs of the reference's QM9 task (QM9Task.py:168-187): Dipole
    (outputs.py:379-468) as the task builds it (magnitude, standardised charges) and in its vector form with a narrower
    hidden width, and ElectronicSpatialExtentV2 (471-545).  `ase` is not installed here: its mass table is replaced by
    the product's built-in table (gotennet_amd.outputs._ATOMIC_MASS), so the masses themselves are NOT pinned by this
    fixture -- everything downstream of them is."""
    sys.path.insert(0, ROOT)
    from gotennet_amd.outputs import _ATOMIC_MASS
    masses = np.zeros(119)
    masses[:len(_ATOMIC_MASS)] = _ATOMIC_MASS
    sys.modules.setdefault("torch_scatter", types.ModuleType("torch_scatter"))
    sys.modules["torch_scatter"].scatter = ref_shims._scatter
    sys.modules.setdefault("ase", types.ModuleType("ase"))
    sys.modules.setdefault("ase.data", types.ModuleType("ase.data"))
    sys.modules["ase"].data = sys.modules["ase.data"]
    sys.modules["ase.data"].atomic_masses = masses
    from gotennet.models.components import outputs as ref_out
    g = torch.Generator().manual_seed(51)
    F_, D, n_mol = 64, 8, 3
    sizes = [9, 1, 14]
    N = sum(sizes)
    z = torch.randint(1, 10, (N,), generator=g)
    batch = torch.repeat_interleave(torch.arange(n_mol), torch.tensor(sizes))
    h = torch.randn((N, F_), generator=g)
    X = torch.randn((N, D, F_), generator=g) * 0.3
    pos = torch.rand((N, 3), generator=g) * 4.0

    class _D(dict):
        __getattr__ = dict.__getitem__
    inp = _D(z=z, batch=batch, pos=pos, representation=h, vector_representation=X)
    out = dict(h=h.numpy(), X=X.numpy(), z=z.numpy(), batch=batch.numpy(), pos=pos.numpy(), n_mol=np.array(n_mol),
               masses=masses.astype(np.float32))
    dip_task = ref_out.Dipole(n_in=F_, predict_magnitude=True, property="property", mean=torch.tensor(0.3),
                              stddev=torch.tensor(1.7))
    dip_vec = ref_out.Dipole(n_in=F_, n_hidden=32, activation=torch.nn.functional.silu, property="dipole")
    ese = ref_out.ElectronicSpatialExtentV2(n_in=F_, property="property", contributions="contrib")
    sys.modules["ase.data"].atomic_masses = np.ones(120)      # what head_kat() expects
    randomise(dip_task, 5100)
    randomise(dip_vec, 5200)
    randomise(ese, 5300)
    ese.atomic_mass.copy_(torch.from_numpy(masses).float())    # randomise() must not touch the mass table
    with torch.no_grad():
        r1, r2, r3 = dip_task(inp), dip_vec(inp), ese(inp)
    out.update(dip_task_y=r1["property"].numpy(), dip_task_yvec=r1["property_vector"].numpy(),
               dip_vec_y=r2["dipole"].numpy(), dip_vec_yvec=r2["dipole_vector"].numpy(),
               ese_y=r3["property"].numpy(), ese_contrib=r3["contrib"].numpy())
    for tag, m in (("dip_task", dip_task), ("dip_vec", dip_vec), ("ese", ese)):
        for k, v in m.state_dict().items():
            out[f"{tag}/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "kat_qm9_heads.npz"), **out)
    print("kat_qm9_heads written:", sorted(k for k in out if "/" in k)[:6], "...", float(r1["property"][0]), float(r3["property"][0]))


if __name__ == "__main__":
    torch.set_num_threads(1)  # deterministic reduction order for the goldens
    only = sys.argv[1:]                    # optional: names of the fixtures to (re)generate
    for name, (hp, spec) in CASES.items():
        if not only or name in only:
            build(name, hp, spec)
    for name, (hp, spec, wseed, hh) in SEEDED.items():
        if not only or name in only:
            build_seeded(name, hp, spec, wseed, hh)
    if "c2_full_forward_seeded" in only:           # minutes of CPU time: only on request
        build_full_forward()
    for name, spec in WORKLOAD_CASES.items():      # tens of seconds and GBs of autograd state: only on request
        if name in only:
            build_workload(name, *spec)
    if not only:
        sh_kat()
    if not only or "kat_sh_l8" in only:
        sh_kat_high()
    if not only or "kat_qm9_heads" in only:
        qm9_heads_kat()
    if not only or "kat_head" in only:
        head_kat()
