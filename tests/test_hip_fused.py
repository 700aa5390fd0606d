"""GPU: gn_message_fused -- the edge projection with scores / segment softmax / message / aggregate / residual as its
epilogue (no [E, (1+M)F] stream on the inference path; reference gotennet.py:406-407, 452-559, 613-640) --
against the reference fixtures, the CPU oracle and the three-kernel sequence it replaces."""
import pytest
import torch

from tests.golden_util import load_case, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rowptr(deg):
    rp = torch.zeros(len(deg) + 1, dtype=torch.int32)
    rp[1:] = torch.tensor(deg, dtype=torch.int64).cumsum(0).to(torch.int32)
    return rp


@pytest.mark.parametrize("kind", ["uniform20", "ragged", "isolated", "long", "single", "many_empty"])
def test_edge_tiles_partition_the_targets(kind):
    """gn_edge_tiles: tiles are consecutive target ranges that partition [0, N); a tile holds <= 128 rows and <= 128
    targets unless it is ONE target with more rows; the count stays under the host-side bound."""
    from gotennet_amd._lib import call, load, ptr
    g = torch.Generator().manual_seed(1)
    deg = {"uniform20": [20] * 1000,
           "ragged": torch.randint(0, 60, (3000,), generator=g).tolist(),
           "isolated": [0] * 700,
           "long": [5, 300, 7, 128, 129, 1, 0, 0, 640] + [33] * 50,
           "single": [17],
           "many_empty": ([0] * 300 + [128]) * 3}[kind]
    N, E = len(deg), int(sum(deg))
    rp = _rowptr(deg).cuda()
    cap = int(load().gn_edge_tiles_cap(N, E))
    tf = torch.full((cap + 1,), -7, dtype=torch.int32, device="cuda")
    nt = torch.zeros(1, dtype=torch.int32, device="cuda")
    call("gn_edge_tiles", ptr(rp), N, cap, ptr(tf), ptr(nt), None)
    torch.cuda.synchronize()
    n = int(nt.item())
    assert 0 < n <= cap
    first = tf[: n + 1].cpu().tolist()
    assert first[0] == 0 and first[-1] == N and all(a < b for a, b in zip(first, first[1:]))
    rpc = rp.cpu().tolist()
    for a, b in zip(first, first[1:]):
        rows = rpc[b] - rpc[a]
        assert b - a <= 128
        assert rows <= 128 or b - a == 1
    # greedy: two neighbouring tiles inside a 256-target chunk never fit into one
    for a, b, c in zip(first, first[1:], first[2:]):
        if a // 256 == (c - 1) // 256 and c - a <= 128:
            assert rpc[c] - rpc[a] > 128


def _fused_vs_sequence(net, z, ei, ed, ev):
    """(h, X) from the fused kernel and from the three-kernel sequence of the same model."""
    from gotennet_amd import engine
    net.fuse_message = True                          # opt-in
    assert engine.fused_message_ok(net.config())
    h1, X1 = net(z, ei, ed, ev)
    net.fuse_message = False
    assert not engine.fused_message_ok(net.config())
    h0, X0 = net(z, ei, ed, ev)
    torch.cuda.synchronize()
    return (h1, X1), (h0, X0)


@pytest.mark.parametrize("mode", ["f16x2", "split"])
@pytest.mark.parametrize("name", ["c1_qm9_small_seeded", "c2_model_3mol_seeded", "c2_model_lmax4_1mol_seeded"])
def test_fused_matches_golden_and_sequence(name, mode):
    """BASELINE configs[0] (F = 128, lmax 2) and the configs[1] model (F = 256; lmax 2 and 4) through the fused kernel:
    the reference's (h, X) within 1e-4, and the three-kernel sequence within fp32 re-association (the projection values are
    the same bits in the row-wise bf16 arithmetic; the order of the per-target sums differs)."""
    from tests.test_hip_parity import _net_from_case
    cfg, sd, _, t = load_case(name)
    net = _net_from_case(cfg, sd)
    net.gemm_mode = mode
    args = [t[k].cuda() for k in ("z", "edge_index", "edge_diff", "edge_vec")]
    (h1, X1), (h0, X0) = _fused_vs_sequence(net, *args)
    assert rel_err(h1.cpu(), t["h"]) < TOL and rel_err(X1.cpu(), t["X"]) < TOL
    assert rel_err(h1, h0) < 5e-6 and rel_err(X1, X0) < 5e-6
    e_ref = max(rel_err(t["h"], t["h_f64"]), rel_err(t["X"], t["X_f64"]))
    e_hip = max(rel_err(h1.cpu(), t["h_f64"]), rel_err(X1.cpu(), t["X_f64"]))
    assert e_hip < max(10 * e_ref, 1e-5)
    net.fuse_message = True
    h2, X2 = net(*args)                              # bit-reproducible (fixed-order reductions, no atomics)
    assert torch.equal(h1, h2) and torch.equal(X1, X2)


def _random_graph(n_atoms_per_mol, box, seed, drop_self=(), cutoff=5.0, cap=None):
    from oracle import gotennet_oracle as orc
    g = torch.Generator().manual_seed(seed)
    pos = torch.cat([torch.rand((n, 3), generator=g) * box + 40.0 * b for b, n in enumerate(n_atoms_per_mol)])
    batch = torch.cat([torch.full((n,), b, dtype=torch.long) for b, n in enumerate(n_atoms_per_mol)])
    z = torch.randint(1, 9, (pos.shape[0],), generator=g)
    ei, w, vec = orc.distance(pos, batch, cutoff) if cap is None else orc.distance(pos, batch, cutoff, cap)
    if drop_self:                                    # atoms without their self-loop (alone in a molecule: without any edge)
        keep = ~((ei[0] == ei[1]) & torch.isin(ei[0], torch.tensor(list(drop_self))))
        ei, w, vec = ei[:, keep], w[keep], vec[keep]
    return pos, batch, z, ei, w, vec


CONFIGS = [
    # (F, H, lmax, sep_dir, sep_tensor, scale_edge, atoms per molecule, box, cap)
    dict(F=128, H=8, lmax=1, sep_dir=False, sep_tensor=False, scale_edge=True, mols=[9, 9, 9], box=3.0),
    dict(F=128, H=4, lmax=3, sep_dir=True, sep_tensor=False, scale_edge=False, mols=[12, 5, 1, 21], box=3.5, drop=(17,)),
    dict(F=128, H=16, lmax=4, sep_dir=False, sep_tensor=True, scale_edge=True, mols=[14, 14], box=3.0),
    dict(F=256, H=8, lmax=2, sep_dir=True, sep_tensor=True, scale_edge=False, mols=[150, 3, 30], box=4.0, cap=200),   # in-degree 150 > one tile
    dict(F=128, H=8, lmax=2, sep_dir=True, sep_tensor=True, scale_edge=True, mols=[135, 1], box=3.5, cap=200, drop=(135,)),
    dict(F=512, H=8, lmax=2, sep_dir=True, sep_tensor=True, scale_edge=False, mols=[11, 11], box=3.0),
]


@pytest.mark.parametrize("mode", ["f16x2", "split"])
@pytest.mark.parametrize("ci", range(len(CONFIGS)))
def test_fused_matches_oracle_on_flag_and_shape_families(ci, mode):
    """Flags and shapes the full-width fixtures do not reach: shared direction / tensor gates (one block serves every
    degree), scale_edge, 4 / 16 heads, F = 128 / 512, atoms without incoming edges, a molecule of one atom, and targets
    with more incoming edges than a tile holds (chunked path) -- against the CPU oracle and the three-kernel sequence."""
    import gotennet_amd
    from oracle import gotennet_oracle as orc
    c = CONFIGS[ci]
    torch.manual_seed(200 + ci)
    kw = dict(n_atom_basis=c["F"], n_interactions=2, n_rbf=16, num_heads=c["H"], scale_edge=c["scale_edge"], lmax=c["lmax"],
              sep_dir=c["sep_dir"], sep_tensor=c["sep_tensor"])
    net = gotennet_amd.GotenNet(cutoff_fn=gotennet_amd.CosineCutoff(5.0), **kw)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if p.dim() == 1:
                p.uniform_(-0.05, 0.05) if "norm.weight" not in n else p.uniform_(0.9, 1.1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    pos, batch, z, ei, w, vec = _random_graph(c["mols"], c["box"], seed=ci, drop_self=c.get("drop", ()), cap=c.get("cap"))
    deg = torch.bincount(ei[1], minlength=pos.shape[0])
    if c.get("cap"):
        assert int(deg.max()) > 128                  # the chunked path really runs
    if c.get("drop"):
        assert int(deg.min()) == 0                   # an atom without incoming edges
    h_ref, X_ref = orc.gotennet_forward(sd, orc.default_config(**kw), z, ei, w, vec)
    net = net.cuda().eval()
    net.gemm_mode = mode
    net.assume_sorted_edges = True
    (h1, X1), (h0, X0) = _fused_vs_sequence(net, z.cuda(), ei.cuda(), w.cuda(), vec.cuda())
    assert rel_err(h1.cpu(), h_ref) < TOL and rel_err(X1.cpu(), X_ref) < TOL
    assert rel_err(h1, h0) < 1e-5 and rel_err(X1, X0) < 1e-5


def test_fused_falls_back_where_unsupported():
    """F < 128, another activation, lmax > 4, the exact-fp32 arithmetic and the degree-sliced family run the three-kernel
    sequence (engine.fused_message_ok is False) -- nothing is silently approximated."""
    import gotennet_amd
    from gotennet_amd import engine
    mk = lambda **kw: gotennet_amd.GotenNet(cutoff_fn=gotennet_amd.CosineCutoff(5.0), n_interactions=1, n_rbf=8,
                                            **{**dict(n_atom_basis=128, lmax=2), **kw})
    ok = mk()
    ok.gemm_mode = "f16x2"
    assert not engine.fused_message_ok(ok.config())  # opt-in
    ok.fuse_message = True
    assert engine.fused_message_ok(ok.config())
    for bad in (mk(n_atom_basis=64), mk(activation="tanh"), mk(lmax=5), mk(num_heads=64)):
        bad.gemm_mode, bad.fuse_message = "f16x2", True
        assert not engine.fused_message_ok(bad.config())
    ok.gemm_mode = "f32"
    assert not engine.fused_message_ok(ok.config())
    ok.gemm_mode, ok.sliced_kernels = "split", True
    assert not engine.fused_message_ok(ok.config())


def test_energy_only_step_uses_fused_kernel_and_matches_force_step():
    """EnergyForces(forces=False) -- the inference step bench.py reports as `forward_only` -- runs the fused kernel (no
    eproj buffer is allocated) and gives the energies of the energy+force step (which keeps the three-kernel sequence:
    its backward reads eproj)."""
    from tests.test_hip_forces import _head_from_case
    from tests.test_hip_parity import _net_from_case
    from gotennet_amd import _lib
    from gotennet_amd.pipeline import EnergyForces
    cfg, sd, head_sd, t = load_case("c2_model_3mol_seeded")
    net, head = _net_from_case(cfg, sd), _head_from_case(cfg, head_sd)
    args = [t[k].cuda() for k in ("z", "edge_index", "edge_diff", "edge_vec", "batch")] + [cfg["n_mol"]]
    net.fuse_message = True                          # opt-in (off by default: DESIGN.md 5.4)
    ef = EnergyForces(net, head)

    class Seen:
        events, names = [], []

        def want(self, name, a):
            self.names.append(name)
            return None

    _lib.TIMER = Seen()
    try:
        e1, _ = ef(*args, forces=False)
    finally:
        _lib.TIMER = None
    assert "gn_message_fused" in Seen.names and "gn_message_aggregate" not in Seen.names and "gn_attn_softmax" not in Seen.names
    e0, f0 = ef(*args)
    net.fuse_message = False
    e2, _ = ef(*args, forces=False)                  # the default inference path: the three-kernel sequence
    torch.cuda.synchronize()
    assert rel_err(e1.cpu(), e0.cpu()) < 5e-6 and rel_err(e2.cpu(), e0.cpu()) < 5e-6
    assert rel_err(e1.cpu(), t["energy"]) < TOL


# ------------------------------------------------------------------------------------------ EQFF chains as one kernel
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
