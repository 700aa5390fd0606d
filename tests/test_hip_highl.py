"""GPU: degrees 5..8 (gn_highl.hip, the degree-sliced kernels) -- harmonics KAT, and the sliced kernels against the
tuned ones at lmax <= 4.  Model-level parity at lmax = 5..8 (forward, energy, forces against the reference goldens
l5_* .. l8_*) runs with every other fixture in test_hip_parity.py / test_hip_forces.py."""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN_DIR, load_case, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("lmax", [5, 6, 7, 8])
def test_harmonics_known_answers_high_degree(lmax):
    """gn_edge_geometry at l = 5..8 against the reference's TensorInit (layers.py:934-1494) on unit vectors, the axes
    and a self-loop (zero vector, not normalised: gotennet.py:978-980)."""
    from gotennet_amd._lib import call, ptr
    k = np.load(os.path.join(GOLDEN_DIR, "kat_sh_l8.npz"))
    vec = torch.from_numpy(k["vec"])
    unit = torch.cat([vec[:32], vec[40:]])                       # unit vectors, axes, zero
    want = torch.cat([torch.from_numpy(k[f"sh{lmax}"])[:32], torch.from_numpy(k[f"sh{lmax}"])[40:]])
    E, D, R = unit.shape[0], (lmax + 1) ** 2 - 1, 8
    scale = torch.linspace(0.5, 3.0, E, dtype=torch.float64).unsqueeze(1)
    ev = (unit * scale).float().cuda()                           # the kernel normalises: any length must do
    ev[-1] = 0.0
    diff = ev.norm(dim=1)
    src = torch.arange(E, dtype=torch.int32, device="cuda")
    dst = (src + 1) % E
    dst[-1] = src[-1]                                            # the zero vector is a self-loop
    means, betas = torch.linspace(0, 1, R).cuda(), torch.ones(R).cuda()
    rl, phi, cut = (torch.empty(E, D, device="cuda"), torch.empty(E, R, device="cuda"), torch.empty(E, device="cuda"))
    call("gn_edge_geometry", ptr(ev), ptr(diff), ptr(src), ptr(dst), E, lmax, R, 0, ptr(means), ptr(betas), 5.0,
         ptr(rl), ptr(phi), ptr(cut), None)
    torch.cuda.synchronize()
    assert rel_err(rl.cpu().double(), want) < 2e-6
    assert float(rl[-1].abs().max()) == 0.0


# every flag family the sliced kernels branch on: sep / no-sep gates, scale_edge, joint HTR, norej, gamma_w gates,
# TensorLayerNorm, composed edge updates (direct HTR backward)
_CASES = ["l1_nosep_scale_f32", "l2_sep_f32", "l2_mixed_f64ch", "l3_sep_scale_f32", "l4_sep_f32", "opt_bessel_norej",
          "opt_gauss_jointhtr_gated", "opt_jointhtr_l3_tanh", "opt_act_norej_joint", "opt_tln_l4",
          "opt_mlp_linwa_ln_gated", "c2_model_lmax4_1mol_seeded"]


@pytest.mark.parametrize("name", _CASES)
def test_degree_sliced_kernels_match_tuned_kernels(name):
    """``net.sliced_kernels = True`` ORs GN_LMAX_SLICED into the lmax argument of the message / HTR entry points (an
    explicit per-call flag: the library reads no environment variable) and routes lmax <= 4 through gn_highl.hip: the
    representation must come out the same as from the tuned kernels (message stage: same per-row arithmetic; HTR
    differs in the literal-vs-closed rejection form only for the default mode), forces within 1e-5 -- in ONE process,
    both families side by side."""
    from tests.test_hip_forces import _head_from_case
    from tests.test_hip_parity import _net_from_case
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case(name)
    net, head = _net_from_case(cfg, sd), _head_from_case(cfg, head_sd)
    args = [t[k].cuda() for k in ("z", "edge_index", "edge_diff", "edge_vec")]
    outs = {}
    for flag in (False, True):
        net.sliced_kernels = flag
        assert net.config().sliced == flag
        h, X = net(*args)
        e, f = EnergyForces(net, head)(*args, t["batch"].cuda(), cfg["n_mol"])
        torch.cuda.synchronize()
        outs[flag] = [v.cpu() for v in (h, X, e, f)]
    for what, a, b in zip("hXef", outs[False], outs[True]):
        assert rel_err(b, a) < 1e-5, (name, what, rel_err(b, a))
    # and the sliced kernels on their own against the reference
    h, X, e, f = outs[True]
    assert rel_err(h, t["h"]) < 1e-4 and rel_err(X, t["X"]) < 1e-4, name
    assert rel_err(e, t["energy"]) < 1e-4 and rel_err(f, t["forces"]) < 1e-4, name


@pytest.mark.parametrize("lmax,sep,F", [(5, True, 128), (6, False, 256), (8, True, 64)])
def test_high_degree_wide_model_matches_oracle(lmax, sep, F):
    """Wider features (1, 2 and 4 slots per workgroup row) at lmax 5 / 6 / 8 on a 2-molecule batch against the CPU
    oracle, incl. forces, bit-reproducibility and zero net force per molecule."""
    import gotennet_amd
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    from oracle import gotennet_oracle as orc
    from tests.test_hip_parity import _synthetic
    torch.manual_seed(100 + lmax)
    kw = dict(n_atom_basis=F, n_interactions=2, n_rbf=16, num_heads=8, scale_edge=not sep, lmax=lmax, sep_dir=sep,
              sep_tensor=sep)
    net = gotennet_amd.GotenNet(cutoff_fn=gotennet_amd.CosineCutoff(5.0), **kw)
    head = Atomwise(n_in=F, n_hidden=32, derivative="forces", activation="silu")
    with torch.no_grad():
        for m in (net, head):
            for n, p in m.named_parameters():
                if p.dim() == 1:
                    p.uniform_(-0.05, 0.05) if "norm.weight" not in n else p.uniform_(0.9, 1.1)
    sd = {k: v.clone().double() for k, v in net.state_dict().items()}
    hsd = {k: v.clone().double() for k, v in head.state_dict().items()}
    cfg = orc.default_config(**kw)
    pos, batch, z = _synthetic(2, 9, 3.0, seed=lmax)
    e_ref, f_ref, _ = orc.energy_and_forces(sd, cfg, hsd, z, pos.double(), batch, 2)
    net, head = net.cuda().eval(), head.cuda().eval()
    ei, w, vec = distance(pos.cuda(), batch.cuda(), 5.0, 32)
    ef = EnergyForces(net, head)
    e, f = ef(z.cuda(), ei, w, vec, batch.cuda(), 2)
    e2, f2 = ef(z.cuda(), ei, w, vec, batch.cuda(), 2)
    torch.cuda.synchronize()
    assert torch.equal(e, e2) and torch.equal(f, f2)
    assert rel_err(e.cpu(), e_ref) < 1e-4
    assert rel_err(f.cpu(), f_ref) < 1e-4
    assert float(f.cpu().reshape(2, 9, 3).sum(1).abs().max()) < 1e-3 * float(f.abs().max())
