"""GPU: the fused EQFF kernels (gn_eqff_fused_forward / _backward: the node-local chain of gotennet.py:716-748 and its
input-gradient as one kernel each way) against the reference fixtures and the launch sequence they replace."""
import pytest
import torch

from tests.golden_util import load_case, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("mode", ["f16x2", "split"])
@pytest.mark.parametrize("name", ["c1_qm9_small_seeded", "c2_model_3mol_seeded", "c2_model_lmax4_1mol_seeded"])
def test_eqff_fused_kernels_match_golden_and_sequence(name, mode):
    """gn_eqff_fused_forward / _backward (context -> gamma_m.0 -> gamma_m.1 -> update, and the input-gradient chain, one
    kernel each; reference gotennet.py:716-748) on the full-width fixtures (F = 128 and 256; N = 19 / 63 / 21: not a
    multiple of the 16-atom tile): (h, X), energies and forces against the reference, and against the launch sequence
    (`fuse_eqff = False`) within fp32 re-association."""
    from tests.test_hip_forces import _head_from_case
    from tests.test_hip_parity import _net_from_case
    from gotennet_amd import engine
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case(name)
    net, head = _net_from_case(cfg, sd), _head_from_case(cfg, head_sd)
    net.gemm_mode = mode
    args = [t[k].cuda() for k in ("z", "edge_index", "edge_diff", "edge_vec")]
    out = {}
    for fused in (True, False):
        net.fuse_eqff = fused
        assert engine.eqff_fused_ok(net.config()) == fused
        h, X = net(*args)
        e, f = EnergyForces(net, head)(*args, t["batch"].cuda(), cfg["n_mol"])
        torch.cuda.synchronize()
        out[fused] = [v.cpu() for v in (h, X, e, f)]
    h, X, e, f = out[True]
    assert rel_err(h, t["h"]) < TOL and rel_err(X, t["X"]) < TOL
    assert rel_err(e, t["energy"]) < TOL and rel_err(f, t["forces"]) < TOL
    for a, b in zip(out[True], out[False]):
        assert rel_err(a, b) < 1e-5
    if mode == "split":
        # row-wise arithmetic, same k order, same term order, same row order in the element-wise sums: the fused chain gives
        # the launch sequence's BITS, so the auto switch (fused up to 1 024 atoms per call) cannot break the bit-exact batch
        # independence this arithmetic promises
        for a, b in zip(out[True], out[False]):
            assert torch.equal(a, b)
    net.fuse_eqff = True
    e2, f2 = EnergyForces(net, head)(*args, t["batch"].cuda(), cfg["n_mol"])
    assert torch.equal(e2.cpu(), e) and torch.equal(f2.cpu(), f)          # bit-reproducible


def test_eqff_fused_falls_back_where_unsupported():
    import gotennet_amd
    from gotennet_amd import engine
    mk = lambda **kw: gotennet_amd.GotenNet(cutoff_fn=gotennet_amd.CosineCutoff(5.0), n_interactions=1, n_rbf=8,
                                            **{**dict(n_atom_basis=128, lmax=2), **kw})
    ok = mk()
    ok.gemm_mode = "f16x2"
    assert ok.fuse_eqff is None                      # auto: by system size
    assert engine.eqff_fused_ok(ok.config(), 21) and not engine.eqff_fused_ok(ok.config(), 2688)
    assert not engine.eqff_fused_ok(ok.config())     # (size unknown: the launch sequence)
    ok.fuse_eqff = False
    assert not engine.eqff_fused_ok(ok.config(), 21)
    ok.fuse_eqff = True
    assert engine.eqff_fused_ok(ok.config()) and engine.eqff_fused_ok(ok.config(), 2688)
    ok.gemm_mode = "f32"
    assert not engine.eqff_fused_ok(ok.config())
    for bad in (mk(n_atom_basis=64), mk(n_atom_basis=512), mk(activation="tanh")):
        bad.gemm_mode, bad.fuse_eqff = "f16x2", True
        assert not engine.eqff_fused_ok(bad.config())
