"""The column-loop projection kernel (gn_gemm_colpipe.hip: f16x2 groups with K = 256, a product >= 1024 columns wide and
>= 2048 output tiles of 64 x 128, no prologue) and the edge-sized shapes around it (which keep the slab kernel) against
fp64 products: ragged M / N, every epilogue, row maps, K-segmented A, riders, per-row exponents."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def _r(g):
    return lambda *s: torch.randn(*s, device="cuda", generator=g)


silu = torch.nn.functional.silu


@pytest.mark.parametrize("M,N,K", [(33000, 256, 256), (11000, 1536, 256), (17013, 1032, 256), (19000, 1000, 256), (40000, 200, 256),
                                   (33000, 256, 512), (33000, 256, 1536), (17001, 520, 768)])
def test_colpipe_and_edge_sized_epilogues(M, N, K):
    from gotennet_amd import engine
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    r = _r(g)
    dd = lambda t: t.double()
    A, W, b = r(M, K), r(N, K) / 8, r(N)
    C, P = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    lo, hi = (N // 16) * 4, (N // 8) * 4
    engine.gemm(A, K, W, b, C, N, M, N, K, act=(lo, hi), pre_out=P, mode="f16x2")
    pre = dd(A) @ dd(W).T + dd(b)
    ref = pre.clone(); ref[:, lo:hi] = silu(ref[:, lo:hi])
    assert rel_err(P, pre) < 2e-6 and rel_err(C, ref) < 2e-6
    res, gate = r(M, N), r(M, N)
    engine.gemm(A, K, W, b, C, N, M, N, K, act=(0, N), res=res, gate=gate, mode="f16x2")
    assert rel_err(C, dd(res) + silu(pre) * dd(gate)) < 2e-6
    engine.gemm(A, K, W, None, C, N, M, N, K, dgate=gate, mode="f16x2")          # output * SiLU'(gate)
    sg = torch.sigmoid(dd(gate))
    assert rel_err(C, (dd(A) @ dd(W).T) * (sg * (1 + dd(gate) * (1 - sg)))) < 2e-6
    # res == C (in place), as the backward uses it
    C2 = res.clone()
    engine.gemm(A, K, W, None, C2, N, M, N, K, res=C2, mode="f16x2")
    assert rel_err(C2, dd(res) + dd(A) @ dd(W).T) < 2e-6


def test_edge_sized_group_rowmap_segments():
    """A group as the step issues it: the edge-sized product with atom-sized riders of other shapes; row maps; the
    K-segmented A operand in whole chunks; a strided output (ldc > N, column offset)."""
    from gotennet_amd import engine
    g = torch.Generator(device="cuda").manual_seed(5)
    r = _r(g)
    dd = lambda t: t.double()
    E, Na = 30011, 1344
    A0, W0, b0, C0 = r(E, 256), r(1536, 256) / 8, r(1536), torch.empty(E, 1536, device="cuda")
    A1, W1, b1 = r(Na, 256), r(1024, 256) / 8, r(1024)
    C1, P1 = torch.empty(Na, 1024, device="cuda"), torch.empty(Na, 1024, device="cuda")
    engine.gemm_group([dict(A=A0, lda=256, W=W0, bias=b0, C=C0, ldc=1536, rows=E, nout=1536, K=256),
                       dict(A=A1, lda=256, W=W1, bias=b1, C=C1, ldc=1024, rows=Na, nout=1024, K=256, act=(512, 1024), pre_out=P1)],
                      mode="f16x2")
    assert rel_err(C0, dd(A0) @ dd(W0).T + dd(b0)) < 2e-6
    pre = dd(A1) @ dd(W1).T + dd(b1)
    ref = pre.clone(); ref[:, 512:] = silu(ref[:, 512:])
    assert rel_err(P1, pre) < 2e-6 and rel_err(C1, ref) < 2e-6
    # the backward's group: K = 1536 edge-sized + K = 1280 / 1280 / 512 riders writing column blocks of one [Na, 1024] tensor
    G0, Wt0, R0, D0 = r(E, 1536), r(256, 1536) / 8, r(E, 256), torch.empty(E, 256, device="cuda")
    gx, gv, Ws, Wv = r(Na, 1280), r(Na, 1280), r(256, 1280) / 8, r(256, 1280) / 8
    gn, pre_n = torch.zeros(Na, 1024, device="cuda"), r(Na, 1024)
    gq, Wqk, Rq, Dq = r(Na, 512), r(256, 512) / 8, r(Na, 256), torch.empty(Na, 256, device="cuda")
    engine.gemm_group([dict(A=G0, lda=1536, W=Wt0, C=D0, ldc=256, rows=E, nout=256, K=1536, res=R0),
                       dict(A=gx, lda=1280, W=Ws, C=gn, ldc=1024, rows=Na, nout=256, K=1280, c_off=512, dgate=pre_n, g_off=512),
                       dict(A=gv, lda=1280, W=Wv, C=gn, ldc=1024, rows=Na, nout=256, K=1280, c_off=768, dgate=pre_n, g_off=768),
                       dict(A=gq, lda=512, W=Wqk, C=Dq, ldc=256, rows=Na, nout=256, K=512, res=Rq)], mode="f16x2")
    assert rel_err(D0, dd(R0) + dd(G0) @ dd(Wt0).T) < 2e-6
    ds = lambda x: torch.sigmoid(x) * (1 + x * (1 - torch.sigmoid(x)))
    assert rel_err(gn[:, 512:768], (dd(gx) @ dd(Ws).T) * ds(dd(pre_n[:, 512:768]))) < 2e-6
    assert rel_err(gn[:, 768:], (dd(gv) @ dd(Wv).T) * ds(dd(pre_n[:, 768:]))) < 2e-6
    assert float(gn[:, :512].abs().max()) == 0.0
    assert rel_err(Dq, dd(Rq) + dd(gq) @ dd(Wqk).T) < 2e-6
    # row maps + K-segmented A (the X-gradient group of the backward): degree blocks of an [n, D, F] tensor
    n, D, Fd = 4100, 8, 256
    X0, X1, X2 = r(n, D, Fd), r(n, D, Fd) * 1e3, r(n, D, Fd) * 1e-3
    Wc, Rr, Cc = r(Fd, 3 * Fd) / 8, r(n, D, Fd), torch.zeros(n, D, Fd, device="cuda")
    engine.gemm_group([dict(A=X0, A2=X1, A3=X2, a_seg=Fd, lda=Fd, W=Wc, C=Cc, ldc=Fd, rows=n * 3, nout=Fd, K=3 * Fd,
                            rowmap=(3, D, 0), res=Rr),
                       dict(A=X0, A2=X1, A3=X2, a_seg=Fd, lda=Fd, W=Wc, C=Cc, ldc=Fd, rows=n * 5, nout=Fd, K=3 * Fd,
                            rowmap=(5, D, 3), res=Rr)], mode="f16x2")
    ref = dd(Rr) + dd(X0) @ dd(Wc)[:, :Fd].T + dd(X1) @ dd(Wc)[:, Fd:2 * Fd].T + dd(X2) @ dd(Wc)[:, 2 * Fd:].T
    assert rel_err(Cc, ref) < 2e-6
    # the forward's X group: one input, four weights, row-mapped outputs
    Ws4 = [r(Fd, Fd) / 8 for _ in range(4)]
    outs = [torch.zeros(n, D, Fd, device="cuda") for _ in range(3)]
    engine.gemm_group([dict(A=X0, lda=Fd, W=Ws4[0], C=outs[0], ldc=Fd, rows=n * D, nout=Fd, K=Fd),
                       dict(A=X0, lda=Fd, W=Ws4[1], C=outs[1], ldc=Fd, rows=n * D, nout=Fd, K=Fd),
                       dict(A=X0, lda=Fd, W=Ws4[2], C=outs[2], ldc=Fd, rows=n * 3, nout=Fd, K=Fd, rowmap=(3, D, 0)),
                       dict(A=X0, lda=Fd, W=Ws4[3], C=outs[2], ldc=Fd, rows=n * 5, nout=Fd, K=Fd, rowmap=(5, D, 3))], mode="f16x2")
    assert rel_err(outs[0], dd(X0) @ dd(Ws4[0]).T) < 2e-6 and rel_err(outs[1], dd(X0) @ dd(Ws4[1]).T) < 2e-6
    ref2 = torch.cat([dd(X0)[:, :3] @ dd(Ws4[2]).T, dd(X0)[:, 3:] @ dd(Ws4[3]).T], 1)
    assert rel_err(outs[2], ref2) < 2e-6


def test_colpipe_row_exponents_hostile_operands():
    """Row-wise exponents of the column-loop kernel (shapes above its thresholds): every row keeps 22 significand bits of ITS
    OWN scale whatever its neighbours hold -- rows ten decades apart, zero rows, non-finite inputs (only their own rows turn
    non-finite), and identical rows give identical bits wherever they sit."""
    from gotennet_amd import engine
    g = torch.Generator(device="cuda").manual_seed(9)
    r = _r(g)
    for (M, N, K) in ((17000, 1024, 256), (11000, 1536, 256), (9100, 2560, 256)):
        W = r(N, K) * 0.1
        dec = torch.randint(-10, 1, (M, 1), device="cuda", generator=g).float()
        A = r(M, K) * (10.0 ** dec)
        A[100:164] = 0.0
        C = torch.empty(M, N, device="cuda")
        run = lambda a: engine.gemm_group([dict(A=a, lda=K, W=W, C=C, ldc=N, rows=a.shape[0], nout=N, K=K)], mode="f16x2")
        run(A)
        ref = A.double() @ W.double().t()
        den = ref.abs().amax(1)
        err = ((C.double() - ref).abs().amax(1) / torch.where(den > 0, den, torch.ones_like(den)))
        assert float(err.max()) < 4e-6, float(err.max())                 # PER ROW, of the row's own max-norm
        assert float(C[100:164].abs().max()) == 0.0
        # batch-position invariance: the same rows at other positions, beside other rows -> the same bits
        perm = torch.randperm(M, device="cuda", generator=g)
        C1 = C.clone()
        run(A[perm].contiguous())
        assert torch.equal(C[:M], C1[perm])
        bad = A.clone()
        bad[7, 5], bad[200, 100] = float("inf"), float("nan")
        run(bad)
        assert not torch.isfinite(C[7]).any() and torch.isnan(C[200]).all()
        assert torch.isfinite(C[6]).all() and torch.isfinite(C[8]).all() and torch.isfinite(C[201]).all()
