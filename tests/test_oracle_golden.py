"""CPU: the oracle (oracle/gotennet_oracle.py) against the reference's golden vectors."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import gotennet_oracle as orc
from tests.golden_util import GOLDEN_DIR, case_names, load_case, rel_err

TOL32 = 2e-5   # fp32 oracle vs fp32 reference (different summation order only)
TOL64 = 1e-11  # fp64 oracle vs fp64 reference


@pytest.mark.parametrize("name", case_names())
def test_forward_fp32_matches_reference(name):
    cfg, sd, _, t = load_case(name)
    h, X, tr = orc.gotennet_forward(sd, cfg, t["z"], t["edge_index"], t["edge_diff"], t["edge_vec"], return_trace=True)
    if "phi" in t:                                   # (the seeded full-width fixtures store final outputs only)
        assert rel_err(tr["phi"], t["phi"]) < 1e-6
        assert rel_err(tr["rl"], t["rl"]) < 1e-6
    for li, (lh, lX, lt) in enumerate(tr["layers"] if "layer0/h" in t else []):
        assert rel_err(lh, t[f"layer{li}/h"]) < TOL32, (li, "h")
        assert rel_err(lX, t[f"layer{li}/X"]) < TOL32, (li, "X")
        assert rel_err(lt, t[f"layer{li}/t"]) < TOL32, (li, "t")
    assert rel_err(h, t["h"]) < TOL32
    assert rel_err(X, t["X"]) < TOL32


@pytest.mark.parametrize("name", case_names())
def test_forward_fp64_matches_reference(name):
    cfg, sd, _, t = load_case(name, torch.float64)
    h, X = orc.gotennet_forward(sd, cfg, t["z"], t["edge_index"], t["edge_diff"].double(), t["edge_vec"].double())
    assert rel_err(h, t["h_f64"]) < TOL64
    assert rel_err(X, t["X_f64"]) < TOL64


@pytest.mark.parametrize("name", [n for n in case_names() if "shuffled" not in n])
def test_graph_energy_forces(name):
    cfg, sd, head, t = load_case(name)
    ei, w, vec = orc.distance(t["pos"], t["batch"], cfg["cutoff"])
    assert torch.equal(ei, t["edge_index"])          # bit-exact edge_index
    assert torch.equal(w, t["edge_diff"])
    assert torch.equal(vec, t["edge_vec"])
    e, f, _ = orc.energy_and_forces(sd, cfg, head, t["z"], t["pos"], t["batch"], cfg["n_mol"])
    assert rel_err(e, t["energy"]) < TOL32
    assert rel_err(f, t["forces"]) < 5e-5
    cfg, sd, head, t = load_case(name, torch.float64)
    e, f, _ = orc.energy_and_forces(sd, cfg, head, t["z"], t["pos"].double(), t["batch"], cfg["n_mol"])
    assert rel_err(e, t["energy_f64"]) < TOL64
    assert rel_err(f, t["forces_f64"]) < 1e-10


def test_basis_known_answers():
    k = np.load(os.path.join(GOLDEN_DIR, "kat_basis.npz"))
    u = torch.from_numpy(k["unit"])
    for l in (1, 2, 3, 4):
        got = orc.real_harmonics(l, u)
        assert torch.allclose(got, torch.from_numpy(k[f"sh{l}"]), rtol=0, atol=1e-14)
    d = torch.from_numpy(k["d"])
    assert torch.allclose(orc.cosine_cutoff(d, 5.0), torch.from_numpy(k["cut"]), rtol=0, atol=1e-15)
    for R in (8, 32):
        means, betas = orc.expnorm_params(5.0, R)
        assert torch.equal(means, torch.from_numpy(k[f"rbf{R}_means"]))
        assert torch.equal(betas, torch.from_numpy(k[f"rbf{R}_betas"]))
        got = orc.expnorm_smearing(d, means.double(), betas.double(), 5.0)
        assert torch.allclose(got, torch.from_numpy(k[f"rbf{R}"]), rtol=0, atol=1e-14)
    # d = 0 (self-loop): C = 1, phi_k = exp(-beta (1 - mu_k)^2); d >= cutoff: 0
    assert float(orc.cosine_cutoff(d, 5.0)[0]) == 1.0
    assert float(orc.cosine_cutoff(d, 5.0)[-2]) == 0.0


def test_high_degree_harmonics_known_answers():
    """Degrees 5..8: the oracle's derived coupling tables against the reference's own formulas (layers.py:934-1494) on
    unit, axis, zero and NON-unit vectors; the same derivation reproduces the hand-restated degree-4 constants."""
    k = np.load(os.path.join(GOLDEN_DIR, "kat_sh_l8.npz"))
    v = torch.from_numpy(k["vec"])
    for l in (5, 6, 7, 8):
        got = orc.real_harmonics(l, v)
        want = torch.from_numpy(k[f"sh{l}"])
        assert got.shape == want.shape == (v.shape[0], (l + 1) ** 2 - 1)
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-13), l
    assert float(orc.real_harmonics(8, v)[-1].abs().max()) == 0.0          # zero vector (self-loop direction)
    T4 = orc.harmonic_raise_table(4)
    assert abs(T4[0, 0, 2] - 0.75 * math.sqrt(2)) < 1e-12 and abs(T4[4, 3, 1] - 3 / 7 * math.sqrt(7)) < 1e-12
    assert abs(T4[2, 2, 2] - 3 / 56 * math.sqrt(210)) < 1e-12 and np.count_nonzero(T4) == 31
    # the reference's degrees >= 3 are not rotation-covariant (degree 3 mixes two normalisations): documented quirk
    u = v[:32]
    n3 = orc.real_harmonics(3, u)[:, 8:15].pow(2).sum(1)
    assert float(n3.max() - n3.min()) > 1.0


def test_segment_softmax_single_edge_and_empty():
    s = torch.tensor([[3.0], [1.0], [2.0], [7.0]])
    idx = torch.tensor([0, 0, 0, 2])               # node 1 has no edges, node 2 a single edge
    a = orc.segment_softmax(s, idx, 3)
    assert abs(float(a[:3].sum()) - 1.0) < 1e-6
    assert abs(float(a[3]) - 1.0) < 1e-6
