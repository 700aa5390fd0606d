"""GPU: the HIP path (through the C ABI) against the reference's golden vectors and the oracle."""
import pytest
import torch

from tests.golden_util import case_names, load_case, rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]

# north_star tolerance: (h, X) within 1e-4 relative (max-norm per tensor, fp32)
TOL = 1e-4


def _net_from_case(cfg, sd):
    import gotennet_amd
    net = gotennet_amd.GotenNet(
        n_atom_basis=cfg["n_atom_basis"], n_interactions=cfg["n_interactions"], n_rbf=cfg["n_rbf"],
        cutoff_fn=gotennet_amd.CosineCutoff(cfg["cutoff"]), max_z=cfg["max_z"], num_heads=cfg["num_heads"],
        scale_edge=cfg["scale_edge"], lmax=cfg["lmax"], sep_dir=cfg["sep_dir"], sep_tensor=cfg["sep_tensor"],
        # non-default flags carried by the opt_* fixtures
        sep_htr=cfg.get("sep_htr", True), radial_basis=cfg.get("radial_basis", "expnorm"),
        edge_updates=cfg.get("edge_updates", True), layernorm=cfg.get("layernorm", ""),
        steerable_norm=cfg.get("steerable_norm", ""), edge_ln=cfg.get("edge_ln", ""),
            activation=cfg.get("activation", "silu"), evec_dim=cfg.get("evec_dim"), emlp_dim=cfg.get("emlp_dim"),
        aggr=cfg.get("aggr", "add"))
    net.load_state_dict(sd, strict=True)
    return net.cuda().eval()


@pytest.mark.parametrize("name", case_names())
def test_forward_matches_golden(name):
    cfg, sd, _, t = load_case(name)
    net = _net_from_case(cfg, sd)
    ev = t["edge_vec"].cuda()
    ev0 = ev.clone()
    trace = []
    h, X = net(t["z"].cuda(), t["edge_index"].cuda(), t["edge_diff"].cuda(), ev, _trace=trace)
    torch.cuda.synchronize()
    assert torch.equal(ev, ev0)                                  # inputs untouched
    assert h.shape == t["h"].shape and X.shape == t["X"].shape
    if "shuffled" not in name and "layer0/h" in t:               # per-layer t is in CSR order
        for li, (lh, lX, lt) in enumerate(trace):
            assert rel_err(lh.cpu(), t[f"layer{li}/h"]) < TOL, (li, "h")
            assert rel_err(lX.cpu(), t[f"layer{li}/X"]) < TOL, (li, "X")
            assert rel_err(lt.cpu(), t[f"layer{li}/t"]) < TOL, (li, "t")
    assert rel_err(h.cpu(), t["h"]) < TOL
    assert rel_err(X.cpu(), t["X"]) < TOL
    # and against the fp64 truth: the HIP path must not be meaningfully worse than the fp32 reference
    e_ref = max(rel_err(t["h"], t["h_f64"]), rel_err(t["X"], t["X_f64"]))
    e_hip = max(rel_err(h.cpu(), t["h_f64"]), rel_err(X.cpu(), t["X_f64"]))
    assert e_hip < max(10 * e_ref, 1e-5)


@pytest.mark.parametrize("name", ["l2_sep_f32", "l3_sep_scale_f32"])
def test_edge_basis_matches_golden(name):
    from gotennet_amd import engine
    cfg, sd, _, t = load_case(name)
    net = _net_from_case(cfg, sd)
    g = engine.Graph(net.config(), net.packed_weights(), t["z"].shape[0], t["edge_index"].cuda(),
                     t["edge_diff"].cuda(), t["edge_vec"].cuda())
    torch.cuda.synchronize()
    assert rel_err(g.rl.cpu(), t["rl"]) < 1e-6
    assert rel_err(g.phi.cpu(), t["phi"]) < 1e-6
    ei = t["edge_index"]
    assert torch.equal(g.src.cpu().long(), ei[0]) and torch.equal(g.dst.cpu().long(), ei[1])
    N = t["z"].shape[0]
    rp = torch.zeros(N + 1, dtype=torch.long)
    rp[1:] = torch.bincount(ei[1], minlength=N).cumsum(0)
    assert torch.equal(g.rowptr.cpu().long(), rp)                # bit-exact index work


@pytest.mark.parametrize("name", [n for n in case_names() if "shuffled" not in n])
def test_radius_graph_bit_exact(name):
    from gotennet_amd.graph import distance
    cfg, _, _, t = load_case(name)
    ei, w, vec = distance(t["pos"].cuda(), t["batch"].cuda(), cfg["cutoff"], 32)
    assert torch.equal(ei.cpu(), t["edge_index"])                # edge_index bit-exact
    assert torch.equal(vec.cpu(), t["edge_vec"])
    assert rel_err(w.cpu(), t["edge_diff"]) < 1e-6


def test_radius_graph_neighbor_cap_and_oracle():
    from gotennet_amd.graph import distance
    from oracle import gotennet_oracle as orc
    g = torch.Generator().manual_seed(3)
    pos = torch.rand((90, 3), generator=g) * 4.0
    batch = torch.cat([torch.zeros(50, dtype=torch.long), torch.ones(40, dtype=torch.long)])
    for cap in (8, 32, 64):
        ei, w, vec = distance(pos.cuda(), batch.cuda(), 5.0, cap)
        ref = orc.radius_graph(pos, batch, 5.0, cap, loop=True)
        assert torch.equal(ei.cpu(), ref)


def test_wrapper_matches_golden():
    import types
    import gotennet_amd
    cfg, sd, _, t = load_case("l2_sep_f32")
    net = gotennet_amd.GotenNetWrapper(
        n_atom_basis=cfg["n_atom_basis"], n_interactions=cfg["n_interactions"], n_rbf=cfg["n_rbf"],
        cutoff_fn=gotennet_amd.CosineCutoff(cfg["cutoff"]), max_z=cfg["max_z"], num_heads=cfg["num_heads"],
        scale_edge=cfg["scale_edge"], lmax=cfg["lmax"], sep_dir=cfg["sep_dir"], sep_tensor=cfg["sep_tensor"],
        # non-default flags carried by the opt_* fixtures
        sep_htr=cfg.get("sep_htr", True), radial_basis=cfg.get("radial_basis", "expnorm"),
        edge_updates=cfg.get("edge_updates", True), layernorm=cfg.get("layernorm", ""),
        steerable_norm=cfg.get("steerable_norm", ""), edge_ln=cfg.get("edge_ln", ""),
            activation=cfg.get("activation", "silu"), evec_dim=cfg.get("evec_dim"), emlp_dim=cfg.get("emlp_dim"))
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    inp = types.SimpleNamespace(z=t["z"].cuda(), pos=t["pos"].cuda(), batch=t["batch"].cuda())
    h, X = net(inp)
    assert rel_err(h.cpu(), t["h"]) < TOL and rel_err(X.cpu(), t["X"]) < TOL


def _synthetic(n_mol, n_atoms, box, seed=0):
    g = torch.Generator().manual_seed(seed)
    pos = torch.cat([torch.rand((n_atoms, 3), generator=g) * box + 20.0 * b for b in range(n_mol)])
    batch = torch.arange(n_mol).repeat_interleave(n_atoms)
    z = torch.randint(1, 9, (n_mol * n_atoms,), generator=g)
    return pos, batch, z


@pytest.mark.parametrize("F,L,lmax,H", [(64, 2, 2, 8), (128, 3, 2, 8), (256, 2, 4, 8), (256, 2, 1, 8)])
def test_forward_matches_oracle_wide(F, L, lmax, H):
    """Wider features than the goldens hold: HIP vs the (golden-pinned) oracle on seeded inputs."""
    import gotennet_amd
    from oracle import gotennet_oracle as orc
    torch.manual_seed(F + lmax)
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=L, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=H, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if p.dim() == 1:
                p.uniform_(-0.05, 0.05) if "norm.weight" not in n else p.uniform_(0.9, 1.1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    cfg = orc.default_config(n_atom_basis=F, n_interactions=L, n_rbf=32, num_heads=H, scale_edge=False, lmax=lmax,
                             sep_dir=True, sep_tensor=True)
    pos, batch, z = _synthetic(3, 21, 4.6, seed=F)
    ei, w, vec = orc.distance(pos, batch, 5.0)
    h_ref, X_ref = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
    net = net.cuda().eval()
    h, X = net(z.cuda(), ei.cuda(), w.cuda(), vec.cuda())
    assert rel_err(h.cpu(), h_ref) < TOL
    assert rel_err(X.cpu(), X_ref) < TOL


def test_deterministic_and_empty_rows():
    """Bit-reproducible (no atomics on the float path); atoms without incoming edges keep h, X."""
    import gotennet_amd
    torch.manual_seed(1)
    net = gotennet_amd.GotenNet(n_atom_basis=64, n_interactions=2, n_rbf=16, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                lmax=2, sep_dir=True, sep_tensor=True, scale_edge=True).cuda().eval()
    z = torch.randint(1, 9, (10,)).cuda()
    # atom 9 is isolated and has no self-loop: no incoming edges at all
    src = torch.tensor([0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 4, 3, 4])
    dst = torch.tensor([0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 4])
    ei = torch.stack([src, dst]).cuda()
    pos = torch.rand(10, 3)
    vec = (pos[src] - pos[dst]).cuda()
    w = vec.norm(dim=1)
    a = net(z, ei, w, vec)
    b = net(z, ei, w, vec)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    assert float(a[1][9].abs().max()) == 0.0                     # X of the isolated atom stays zero
    # empty graph
    h, X = net(z, torch.zeros((2, 0), dtype=torch.long).cuda(), torch.zeros(0).cuda(), torch.zeros((0, 3)).cuda())
    assert torch.isfinite(h).all() and float(X.abs().max()) == 0.0


def test_gemm_row_map_and_epilogues():
    from gotennet_amd import engine
    torch.manual_seed(0)
    for (M, N, K) in [(1, 32, 32), (130, 208, 64), (257, 96, 8), (1000, 384, 256)]:
        A = torch.randn(M, K).cuda(); W = torch.randn(N, K).cuda(); b = torch.randn(N).cuda()
        C = torch.empty(M, N).cuda()
        lo, hi = (N // 16) * 4, (N // 8) * 4
        engine.gemm(A, K, W, b, C, N, M, N, K, act=(lo, hi))
        ref = A.double() @ W.double().T + b.double()
        ref[:, lo:hi] = torch.nn.functional.silu(ref[:, lo:hi])
        assert rel_err(C.cpu(), ref.cpu()) < 1e-5
        res = torch.randn(M, N).cuda(); gate = torch.randn(M, N).cuda()
        engine.gemm(A, K, W, b, C, N, M, N, K, act=(0, N), res=res, gate=gate)
        ref = res.double() + torch.nn.functional.silu(A.double() @ W.double().T + b.double()) * gate.double()
        assert rel_err(C.cpu(), ref.cpu()) < 1e-5
    # row map: degree-2 rows (offset 3, count 5) of an [n, D=8, F] tensor
    n, D, Fd = 37, 8, 64
    X = torch.randn(n, D, Fd).cuda(); W = torch.randn(Fd, Fd).cuda()
    out = torch.zeros(n, D, Fd).cuda()
    engine.gemm(X, Fd, W, None, out, Fd, n * 5, Fd, Fd, rowmap=(5, D, 3))
    ref = torch.zeros(n, D, Fd, dtype=torch.double)
    ref[:, 3:8] = X[:, 3:8].double().cpu() @ W.double().cpu().T
    assert rel_err(out.cpu(), ref) < 1e-5
    assert float(out[:, :3].abs().max()) == 0.0


def test_gemm_group_and_segmented_k():
    """gn_gemm_group: unlike problems in one launch (different M/N/K, epilogues, row maps), equal problems (one XCD
    cut), more than four problems (chunked), and the K-segmented A operand -- against fp64 matmuls."""
    from gotennet_amd import engine
    torch.manual_seed(1)
    dev = "cuda"
    r = lambda *s: torch.randn(*s, device=dev)
    # four unlike problems (spread mode): big rows, a rider with a different K, an activated one, a row-mapped one
    A0, W0, b0, C0 = r(3000, 64), r(256, 64), r(256), torch.empty(3000, 256, device=dev)
    A1, W1, C1, R1 = r(130, 512), r(96, 512), torch.empty(130, 96, device=dev), r(130, 96)
    A2, W2, b2, C2, P2 = r(257, 32), r(128, 32), r(128), torch.empty(257, 128, device=dev), torch.empty(257, 128, device=dev)
    n, D, Fd = 29, 8, 64
    X, W3, C3 = r(n, D, Fd), r(Fd, Fd), torch.zeros(n, D, Fd, device=dev)
    engine.gemm_group([
        dict(A=A0, lda=64, W=W0, bias=b0, C=C0, ldc=256, rows=3000, nout=256, K=64),
        dict(A=A1, lda=512, W=W1, C=C1, ldc=96, rows=130, nout=96, K=512, res=R1),
        dict(A=A2, lda=32, W=W2, bias=b2, C=C2, ldc=128, rows=257, nout=128, K=32, act=(0, 64), pre_out=P2),
        dict(A=X, lda=Fd, W=W3, C=C3, ldc=Fd, rows=n * 3, nout=Fd, K=Fd, rowmap=(3, D, 0))])
    dd = lambda t: t.double().cpu()
    assert rel_err(C0.cpu(), dd(A0) @ dd(W0).T + dd(b0)) < 1e-5
    assert rel_err(C1.cpu(), dd(R1) + dd(A1) @ dd(W1).T) < 1e-5
    pre = dd(A2) @ dd(W2).T + dd(b2)
    assert rel_err(P2.cpu(), pre) < 1e-5
    ref2 = pre.clone(); ref2[:, :64] = torch.nn.functional.silu(ref2[:, :64])
    assert rel_err(C2.cpu(), ref2) < 1e-5
    ref3 = torch.zeros(n, D, Fd, dtype=torch.double); ref3[:, :3] = dd(X)[:, :3] @ dd(W3).T
    assert rel_err(C3.cpu(), ref3) < 1e-5 and float(C3[:, 3:].abs().max()) == 0.0
    # six equal problems (one cut of the concatenated tile list; chunks of four)
    As = [r(500, 64) for _ in range(6)]; Ws = [r(192, 64) for _ in range(6)]
    Cs = [torch.empty(500, 192, device=dev) for _ in range(6)]
    engine.gemm_group([dict(A=a, lda=64, W=w, C=c, ldc=192, rows=500, nout=192, K=64) for a, w, c in zip(As, Ws, Cs)])
    for a, w, c in zip(As, Ws, Cs):
        assert rel_err(c.cpu(), dd(a) @ dd(w).T) < 1e-5
    # K-segmented A: C = res + A W_0^T + A2 W_1^T + A3 W_2^T with the weights concatenated along K, row-mapped
    G0, G1, G2 = r(n, D, Fd), r(n, D, Fd), r(n, D, Fd)
    Wc, Rr, Cc = r(Fd, 3 * Fd), r(n, D, Fd), torch.zeros(n, D, Fd, device=dev)
    engine.gemm_group([dict(A=G0, A2=G1, A3=G2, a_seg=Fd, lda=Fd, W=Wc, C=Cc, ldc=Fd, rows=n * 5, nout=Fd, K=3 * Fd,
                            rowmap=(5, D, 3), res=Rr)])
    refc = torch.zeros(n, D, Fd, dtype=torch.double)
    refc[:, 3:] = (dd(Rr) + dd(G0) @ dd(Wc)[:, :Fd].T + dd(G1) @ dd(Wc)[:, Fd:2 * Fd].T + dd(G2) @ dd(Wc)[:, 2 * Fd:].T)[:, 3:]
    assert rel_err(Cc.cpu(), refc) < 1e-5


@pytest.mark.parametrize("K", [128, 256, 512, 768, 1536])
def test_gemm_panel_kernel_paths(K):
    """The K-resident panel kernel (gn_gemm_panel.hip: f16x2 groups of <= 2048 tiles of 32 x 128, no prologue, one depth
    K in {128, 256, 512} -- one panel per tile -- or depths that are multiples of 256 -- 256-deep chunks): ragged M / N, every epilogue, row maps, the K-segmented A operand and a group of equal-K problems,
    against fp64 products; the same problems through the two other arithmetics (slab kernels) for comparison."""
    from gotennet_amd import engine
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(K)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    dd = lambda t: t.double().cpu()
    silu = torch.nn.functional.silu
    for mode in ("f16x2", "split", "f32"):
        for (M, N) in [(1, 32), (63, 64), (130, 208), (257, 96), (1000, 384), (2688, 512)]:
            A, W, b = r(M, K), r(N, K) / 8, r(N)
            C, P = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
            lo, hi = (N // 16) * 4, (N // 8) * 4
            engine.gemm(A, K, W, b, C, N, M, N, K, act=(lo, hi), pre_out=P, mode=mode)
            pre = dd(A) @ dd(W).T + dd(b)
            ref = pre.clone(); ref[:, lo:hi] = silu(ref[:, lo:hi])
            assert rel_err(P.cpu(), pre) < 2e-6 and rel_err(C.cpu(), ref) < 2e-6
            res, gate = r(M, N), r(M, N)
            engine.gemm(A, K, W, b, C, N, M, N, K, act=(0, N), res=res, gate=gate, mode=mode)
            assert rel_err(C.cpu(), dd(res) + silu(pre) * dd(gate)) < 2e-6
            engine.gemm(A, K, W, None, C, N, M, N, K, dgate=gate, mode=mode)          # output * SiLU'(gate)
            sg = torch.sigmoid(dd(gate))
            assert rel_err(C.cpu(), (dd(A) @ dd(W).T) * (sg * (1 + dd(gate) * (1 - sg)))) < 2e-6
        # row map: degree-2 rows (offset 3, count 5) of an [n, D = 8, K] tensor; rows outside stay untouched
        n, D = 37, 8
        X, W = r(n, D, K), r(K, K) / 8
        out = torch.zeros(n, D, K, device=dev)
        engine.gemm(X, K, W, None, out, K, n * 5, K, K, rowmap=(5, D, 3), mode=mode)
        ref = torch.zeros(n, D, K, dtype=torch.double); ref[:, 3:] = dd(X)[:, 3:] @ dd(W).T
        assert rel_err(out.cpu(), ref) < 2e-6 and float(out[:, :3].abs().max()) == 0.0
        # K-segmented A in whole 128-column chunks + residual, and three equal-K problems in one launch
        nseg = 0 if K == 128 else (2 if K <= 512 else 3)
        seg = K // nseg if nseg else 0
        probs, refs = [], []
        for q in range(3):
            M, N = (300, 200, 77)[q], (64, 160, 256)[q]
            W, C = r(N, K) / 8, torch.empty(M, N, device=dev)
            if seg and q == 0:
                As, R = [r(M, seg) * 10.0 ** (2 * i) for i in range(nseg)], r(M, N)      # segments 100x apart
                probs.append(dict(A=As[0], A2=As[1], A3=As[2] if nseg == 3 else None, a_seg=seg, lda=seg, W=W, C=C, ldc=N,
                                  rows=M, nout=N, K=K, res=R))
                refs.append((C, dd(R) + torch.cat([dd(a) for a in As], 1) @ dd(W).T))
            else:
                A = r(M, K)
                probs.append(dict(A=A, lda=K, W=W, C=C, ldc=N, rows=M, nout=N, K=K))
                refs.append((C, dd(A) @ dd(W).T))
        engine.gemm_group(probs, mode=mode)
        for C, ref in refs:
            assert rel_err(C.cpu(), ref) < 2e-6
    if K == 1536:
        # the one-molecule input-gradient group: depths 1536 / 1280 / 1280 / 512 in one launch (256-deep chunks), and a
        # product whose chunks differ by 1e9 in both orders (the accumulator rows follow the growing block exponent)
        Ms, Ns, Ks = (429, 21, 21, 21), (256, 256, 256, 256), (1536, 1280, 1280, 512)
        As = [r(m, k) for m, k in zip(Ms, Ks)]
        As[0][:, :512] *= 1e-9
        As[1][:, 768:] *= 1e-9
        Ws = [r(n, k) / 8 for n, k in zip(Ns, Ks)]
        Rs = [r(m, n) for m, n in zip(Ms, Ns)]
        Cs = [torch.empty(m, n, device=dev) for m, n in zip(Ms, Ns)]
        engine.gemm_group([dict(A=a, lda=k, W=w, C=c, ldc=n, rows=m, nout=n, K=k, res=rr)
                           for a, w, c, rr, m, n, k in zip(As, Ws, Cs, Rs, Ms, Ns, Ks)], mode="f16x2")
        for a, w, c, rr in zip(As, Ws, Cs, Rs):
            assert rel_err(c.cpu(), dd(rr) + dd(a) @ dd(w).T) < 2e-6


@pytest.mark.parametrize("F,H", [(512, 8), (1024, 16), (16, 4)])
def test_forward_feature_width_extremes(F, H):
    """Slot layout corner cases: F = 512/1024 (a slot spans 2/4 waves), F = 16 (64 slots per workgroup)."""
    import gotennet_amd
    from oracle import gotennet_oracle as orc
    torch.manual_seed(F)
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=2, n_rbf=16, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=H, scale_edge=True, lmax=2, sep_dir=True, sep_tensor=True)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    cfg = orc.default_config(n_atom_basis=F, n_interactions=2, n_rbf=16, num_heads=H, scale_edge=True, lmax=2,
                             sep_dir=True, sep_tensor=True)
    pos, batch, z = _synthetic(2, 12, 3.5, seed=F)
    ei, w, vec = orc.distance(pos, batch, 5.0)
    h_ref, X_ref = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
    net = net.cuda().eval()
    h, X = net(z.cuda(), ei.cuda(), w.cuda(), vec.cuda())
    assert rel_err(h.cpu(), h_ref) < TOL
    assert rel_err(X.cpu(), X_ref) < TOL


def test_softmax_high_degree():
    """One target with 150 incoming edges (> 64 lanes, > one wave pass) and many single-edge targets."""
    import gotennet_amd
    from oracle import gotennet_oracle as orc
    torch.manual_seed(3)
    F = 64
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=2, n_rbf=16, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=True, lmax=1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    cfg = orc.default_config(n_atom_basis=F, n_interactions=2, n_rbf=16, num_heads=8, scale_edge=True, lmax=1)
    n = 151
    g = torch.Generator().manual_seed(2)
    pos = torch.rand((n, 3), generator=g) * 2.5
    z = torch.randint(1, 9, (n,), generator=g)
    # star: everyone -> atom 0, plus self-loops; target-sorted
    src = torch.cat([torch.arange(0, n), torch.arange(1, n)])
    dst = torch.cat([torch.zeros(n, dtype=torch.long), torch.arange(1, n)])
    ei = torch.stack([src, dst])
    vec = pos[src] - pos[dst]
    w = torch.where(src != dst, vec.norm(dim=1), torch.zeros(src.numel()))
    h_ref, X_ref = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
    net = net.cuda().eval()
    h, X = net(z.cuda(), ei.cuda(), w.cuda(), vec.cuda())
    assert rel_err(h.cpu(), h_ref) < TOL
    assert rel_err(X.cpu(), X_ref) < TOL


def test_fp16_block_exponent_products_on_hostile_operands():
    """The default projection arithmetic (two fp16 planes, running block exponents) against an fp64 product on operands
    chosen to stress the scaling: magnitudes from 1e-20 to 1e+20, rows eight decades apart inside one 8-row block, a K
    range whose leading slabs are 1e-9 of the trailing ones (the accumulators must be rescaled mid-tile) and the reverse,
    all-zero blocks, a K-segmented A operand whose three segments differ by 1e6, and non-finite inputs (must propagate,
    not turn finite).  Tolerance: 5e-7 of the output's max-norm (an fp32 matmul is at 5e-7 on the same operands)."""
    from gotennet_amd import engine
    old, engine.GEMM_MODE = engine.GEMM_MODE, "f16x2"
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        rn = lambda *s: torch.randn(*s, device="cuda", generator=g)

        def run(A, W, **kw):
            M, K = A.shape[0], W.shape[1]
            C = torch.empty(M, W.shape[0], device="cuda")
            engine.gemm_group([dict(A=A, lda=A.shape[1], W=W, C=C, ldc=W.shape[0], rows=M, nout=W.shape[0], K=K, **kw)])
            torch.cuda.synchronize()
            return C

        def check(A, W, tol=5e-7):
            C, ref = run(A, W), A.double() @ W.double().t()
            assert rel_err(C.double().cpu(), ref.cpu()) < tol, rel_err(C.double().cpu(), ref.cpu())

        for N in (192, 12032):                                    # 44 panel tiles: the K-resident panel kernel; 2068 (> 2048): the slab kernel
            W = rn(N, 256) * 0.1
            tol = 5e-7 if N == 192 else 9e-7                      # the maximum over 60 x more outputs sits higher
            for scale in (1e-20, 1e-6, 1.0, 1e6, 1e20):
                check(rn(700, 256) * scale, W, tol)
                check(rn(700, 256), W * scale, tol)
            rows = rn(700, 256) * (10.0 ** torch.randint(-8, 1, (700, 1), device="cuda", generator=g).float())
            check(rows, W, tol)
            grow = rn(700, 256)
            grow[:, :96] *= 1e-9                                  # small slabs first: exponents grow, accumulators rescale
            check(grow, W, tol)
            shrink = rn(700, 256)
            shrink[:, 160:] *= 1e-9
            check(shrink, W, tol)
            holes = rn(700, 256)
            holes[64:200] = 0.0
            holes[:, 32:64] = 0.0
            check(holes, W, tol)
            check(torch.zeros(130, 256, device="cuda"), W, tol=1.0)   # 0 / 0: just must not produce NaN
            assert float(run(torch.zeros(130, 256, device="cuda"), W).abs().max()) == 0.0
            # non-finite inputs propagate
            bad = rn(700, 256)
            bad[7, 5], bad[200, 100] = float("inf"), float("nan")
            out = run(bad, W)
            assert not torch.isfinite(out[7]).any() or torch.isnan(out[7]).any()
            assert torch.isnan(out[200]).all()
            assert torch.isfinite(out[100]).all()                 # other 8-row blocks are untouched
        # K-segmented A (three tensors along K, gX = gXp W_vu + gEQ W_vq + gEK W_vk): segments 1e6 apart
        A1, A2, A3 = rn(500, 64) * 1e-3, rn(500, 64) * 1e3, rn(500, 64)
        Wc = rn(64, 192) * 0.1
        C = run(A1, Wc, A2=A2, A3=A3, a_seg=64)
        ref = torch.cat([A1, A2, A3], 1).double() @ Wc.double().t()
        assert rel_err(C.double().cpu(), ref.cpu()) < 5e-7
    finally:
        engine.GEMM_MODE = old


def test_fp16_block_exponent_per_row_bound():
    """The default arithmetic scales the activation rows a WAVE stages by one running exponent: rows 8 q .. 8 q + 7 of
    every 32-row MFMA tile of the workgroup tile (wave q of four), i.e. 8 rows of the panel kernel's 32-row tile, 16 rows of a
    64-row tile or 32 rows of a 128-row tile share it.  A row far below its group keeps fewer bits OF ITS OWN scale.  The bound, per row (d = log2 of the
    group's maximum over the row's own maximum): an element keeps 22 significand bits while it sits within 2^18 of the
    group maximum and loses one bit per binade beyond that (the low fp16 plane bottoms out at 2^-24 of the scaled
    group; past 2^40 the row is flushed), i.e.

        max_n |C[r, n] - ref[r, n]|  <=  32 * max(2^-23, 2^(d_r - 41)) * max_n |ref[r, n]|

    (32 ~ 2 sqrt(K) for K = 256: the worst-case growth of K element errors in a dot product of random operands).  Checked
    on rows spread over TEN decades inside their groups, in both tile shapes; rows within 2^18 of their group -- every
    row of a batch of molecules -- are held to 4e-6 of their own max-norm."""
    from gotennet_amd import engine
    if engine.GEMM_MODE != "f16x2":
        pytest.skip("bound of the fp16 block-exponent arithmetic")
    g = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    K = 256
    # 32-row tiles of the panel kernel (one exponent per 8 rows), 64-row tiles of the slab kernel (2210 panel tiles > 2048,
    # 561 big tiles < 900), the 128 x 128 tile (912 >= 900)
    for M, N, BM in ((704, 192, 32), (4160, 2176, 64), (9728, 1536, 128)):
        W = rn(N, K) * 0.1
        dec = torch.randint(-10, 1, (M, 1), device="cuda", generator=g).float()
        A = rn(M, K) * (10.0 ** dec)
        C = torch.empty(M, N, device="cuda")
        engine.gemm_group([dict(A=A, lda=K, W=W, C=C, ldc=N, rows=M, nout=N, K=K)])
        torch.cuda.synchronize()
        ref = A.double() @ W.double().t()
        err = (C.double() - ref).abs().amax(1) / ref.abs().amax(1)
        row_max = A.abs().amax(1).double()
        r = torch.arange(M, device="cuda")
        group = (r // BM) * 4 + (r % 32) // 8                   # (workgroup tile, staging wave)
        grp_max = torch.zeros(int(group.max()) + 1, dtype=torch.float64, device="cuda").scatter_reduce_(
            0, group, row_max, "amax", include_self=True)[group]
        d = torch.log2(grp_max / row_max)
        bound = 32.0 * torch.maximum(torch.full_like(d, 2.0 ** -23), 2.0 ** (d - 41))
        assert bool((err <= bound).all()), (float((err / bound).max()), float(d[(err / bound).argmax()]))
        near = d <= 18
        assert bool(near.any()) and float(err[near].max()) < 4e-6
        # the bound is not vacuous: the rows it loosens are the rows that need it
        assert float(err[d > 28].max()) > float(err[near].max())


def test_small_feature_molecule_in_a_normal_batch():
    """A molecule whose edges all sit just inside the cutoff (cosine cutoff ~1e-4: its messages are four decades below
    its batch-mates') between two ordinary molecules: its rows share 8-row exponent blocks with theirs at the seams.
    Per-MOLECULE energy and forces against the fp64 oracle, each within 1e-4 of that molecule's own max-norm."""
    import gotennet_amd
    from oracle import gotennet_oracle as orc
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    torch.manual_seed(7)
    F, L = 128, 3
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=L, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=False, lmax=2, sep_dir=True, sep_tensor=True)
    head = Atomwise(n_in=F, n_hidden=64, derivative="forces", activation="silu")
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    cfg = orc.default_config(n_atom_basis=F, n_interactions=L, n_rbf=32, num_heads=8, scale_edge=False, lmax=2,
                             sep_dir=True, sep_tensor=True)
    g = torch.Generator().manual_seed(1)
    normal = lambda off: torch.rand((13, 3), generator=g) * 4.0 + off
    # an equilateral triangle of side 4.97 A: every pair inside the 5 A cutoff by 0.03 A, C(r) = 0.5 (cos(pi r / 5) + 1) ~ 9e-5
    s = 4.97
    far = torch.tensor([[0.0, 0.0, 0.0], [s, 0.0, 0.0], [s / 2, s * 3 ** 0.5 / 2, 0.0]]) + 50.0
    pos = torch.cat([normal(0.0), far, normal(100.0)])
    batch = torch.tensor([0] * 13 + [1] * 3 + [2] * 13)
    z = torch.randint(1, 9, (29,), generator=g)
    e64, f64, _ = orc.energy_and_forces({k: v.double() for k, v in sd.items()}, cfg, {k: v.double() for k, v in hsd.items()},
                                        z, pos.double(), batch, 3)
    ei, w, vec = distance(pos.cuda(), batch.cuda(), 5.0, 32)
    e, f = EnergyForces(net.cuda().eval(), head.cuda().eval())(z.cuda(), ei, w, vec, batch.cuda(), 3)
    e, f = e.cpu().double(), f.cpu().double()
    assert float((ei[0] != ei[1]).sum()) > 6                   # the triangle's six directed edges are in the graph
    for m in range(3):
        rows = batch == m
        assert abs(float(e[m] - e64[m])) <= 1e-4 * abs(float(e64[m])), m
        assert float((f[rows] - f64[rows]).abs().max()) <= 1e-4 * float(f64[rows].abs().max()), m
    # the triangle's forces really are small next to its neighbours' (the point of the case)
    assert float(f64[batch == 1].abs().max()) < 1e-2 * float(f64[batch == 0].abs().max())


def test_softmax_beyond_the_lds_capacity():
    """A 300-neighbour target at 8 heads (2400 scores > the 2048 the workgroup keeps in LDS) next to ordinary targets:
    the forward softmax and the backward's head gradients take their global-memory form for that target and the LDS
    form for the others; energy and forces against the oracle."""
    import gotennet_amd
    from oracle import gotennet_oracle as orc
    from gotennet_amd.outputs import Atomwise
    torch.manual_seed(5)
    F = 64
    net = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=2, n_rbf=16, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=8, scale_edge=True, lmax=2, sep_dir=True, sep_tensor=True)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    cfg = orc.default_config(n_atom_basis=F, n_interactions=2, n_rbf=16, num_heads=8, scale_edge=True, lmax=2,
                             sep_dir=True, sep_tensor=True)
    n = 301
    g = torch.Generator().manual_seed(2)
    pos = torch.rand((n, 3), generator=g) * 2.5
    z = torch.randint(1, 9, (n,), generator=g)
    src = torch.cat([torch.arange(0, n), torch.arange(1, n)])          # star: everyone -> atom 0, plus self-loops
    dst = torch.cat([torch.zeros(n, dtype=torch.long), torch.arange(1, n)])
    ei = torch.stack([src, dst])
    vec = (pos[src] - pos[dst]).requires_grad_(True)
    w = torch.where(src != dst, vec.norm(dim=1), torch.zeros(src.numel()))
    h_ref, X_ref = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
    loss_ref = (h_ref ** 2).sum() + (X_ref ** 2).sum()
    (g_ref,) = torch.autograd.grad(loss_ref, vec)
    net = net.cuda().eval()
    vec_d = vec.detach().cuda().requires_grad_(True)
    w_d = torch.where((src != dst).cuda(), vec_d.norm(dim=1), torch.zeros(src.numel(), device="cuda"))
    h, X = net(z.cuda(), ei.cuda(), w_d, vec_d)
    assert rel_err(h.detach().cpu(), h_ref.detach()) < TOL and rel_err(X.detach().cpu(), X_ref.detach()) < TOL
    (g_hip,) = torch.autograd.grad((h ** 2).sum() + (X ** 2).sum(), vec_d)
    real = src != dst                                  # (a self-loop's vector is pos[i] - pos[i]: its gradient never reaches a position)
    assert rel_err(g_hip.cpu()[real], g_ref[real]) < TOL


@pytest.mark.parametrize("F,H,lmax", [(192, 8, 2), (96, 4, 3), (48, 4, 1), (200, 8, 2), (384, 8, 2)])
def test_feature_width_not_a_power_of_two(F, H, lmax):
    """n_atom_basis that is not a power of two (the reference takes any multiple of num_heads, gotennet.py:767-793) runs
    embedded in the next power-of-two width (gotennet_amd/embed.py: zero-padded weights, channels placed head by head, the
    attention scale folded into gamma_v, the NodeInit LayerNorm kept compact): (h, X) against the oracle, and
    energies and forces (384 runs as 512: slots wider than a wave)."""
    import gotennet_amd
    from oracle import gotennet_oracle as orc
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    torch.manual_seed(F)
    kw = dict(n_atom_basis=F, n_interactions=3, n_rbf=16, num_heads=H, scale_edge=(F % 3 == 0), lmax=lmax, sep_dir=True,
              sep_tensor=True)
    net = gotennet_amd.GotenNet(cutoff_fn=gotennet_amd.CosineCutoff(5.0), **kw)
    with torch.no_grad():                                        # biases are zero-initialised: make them count
        for n_, p_ in net.named_parameters():
            if n_.endswith("bias"):
                p_.normal_(0.0, 0.1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    cfg = orc.default_config(**kw)
    pos, batch, z = _synthetic(3, 13, 3.5, seed=F)
    ei, w, vec = orc.distance(pos, batch, 5.0)
    h_ref, X_ref = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
    net = net.cuda().eval()
    c = net.config()
    assert c.F_model == F and c.F >= F and (c.F & (c.F - 1)) == 0
    h, X = net(z.cuda(), ei.cuda(), w.cuda(), vec.cuda())
    assert h.shape == (39, F) and X.shape == (39, (lmax + 1) ** 2 - 1, F)
    assert rel_err(h.cpu(), h_ref) < TOL and rel_err(X.cpu(), X_ref) < TOL
    head = Atomwise(n_in=F, n_hidden=32, derivative="forces", activation="silu")
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    e_ref, f_ref, _ = orc.energy_and_forces(sd, cfg, hsd, z, pos, batch, 3)
    e, f = EnergyForces(net, head.cuda().eval())(z.cuda(), ei.cuda(), w.cuda(), vec.cuda(), batch.cuda(), 3)
    assert rel_err(e.cpu(), e_ref) < TOL and rel_err(f.cpu(), f_ref) < TOL
    # the reference-style call (autograd through the representation, gradients of a functional of h AND X w.r.t. the edge inputs)
    evr, edr = vec.clone().requires_grad_(True), w.clone().requires_grad_(True)
    h2, X2 = orc.gotennet_forward(sd, cfg, z, ei, edr, evr)
    gv_ref, gd_ref = torch.autograd.grad((h2 * h2).sum() + (X2 * X2).sum(), [evr, edr])
    evc, edc = vec.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    hh, XX = net(z.cuda(), ei.cuda(), edc, evc)
    gv, gd = torch.autograd.grad((hh * hh).sum() + (XX * XX).sum(), [evc, edc])
    mask = ei[0] != ei[1]
    assert rel_err(gv.cpu()[mask], gv_ref[mask]) < TOL and rel_err(gd.cpu()[mask], gd_ref[mask]) < TOL


@pytest.mark.gpu
@pytest.mark.usefixtures("gemm_mode")
@pytest.mark.parametrize("F,H,lmax", [(192, 8, 2), (96, 4, 3)])
def test_input_norms_at_a_width_that_is_not_a_power_of_two(F, H, lmax):
    """VERDICT r5 breadth: ``layernorm`` (nn.LayerNorm on h) and ``steerable_norm`` (TensorLayerNorm on X) at the GATA input
    (gotennet.py:397-398, layers.py:1497-1563) with an embedded width: statistics over the REAL channels (compact -> kernel ->
    padded layout).  (h, X), energies and forces against the oracle."""
    import gotennet_amd
    from oracle import gotennet_oracle as orc
    from gotennet_amd.outputs import Atomwise
    from gotennet_amd.pipeline import EnergyForces
    torch.manual_seed(F + lmax)
    kw = dict(n_atom_basis=F, n_interactions=3, n_rbf=16, num_heads=H, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True,
              layernorm="layer", steerable_norm="layer")
    net = gotennet_amd.GotenNet(cutoff_fn=gotennet_amd.CosineCutoff(5.0), **kw)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("bias"):
                p_.normal_(0.0, 0.1)
            if "layernorm.weight" in n_:
                p_.uniform_(0.5, 1.5)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    cfg = orc.default_config(**kw)
    pos, batch, z = _synthetic(3, 13, 3.5, seed=F)
    ei, w, vec = orc.distance(pos, batch, 5.0)
    h_ref, X_ref = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
    net = net.cuda().eval()
    h, X = net(z.cuda(), ei.cuda(), w.cuda(), vec.cuda())
    assert h.shape == (39, F) and rel_err(h.cpu(), h_ref) < TOL and rel_err(X.cpu(), X_ref) < TOL
    head = Atomwise(n_in=F, n_hidden=32, derivative="forces", activation="silu")
    hsd = {k: v.clone() for k, v in head.state_dict().items()}
    e_ref, f_ref, _ = orc.energy_and_forces(sd, cfg, hsd, z, pos, batch, 3)
    e, f = EnergyForces(net, head.cuda().eval())(z.cuda(), ei.cuda(), w.cuda(), vec.cuda(), batch.cuda(), 3)
    assert rel_err(e.cpu(), e_ref) < TOL and rel_err(f.cpu(), f_ref) < TOL
