"""Where are K7 (gn_htr_edge) and gn_htr_backward bound?  The same launches with (a) the real edge list, (b) every edge's
source = its target (gathered rows L1/L2-hot: the gather costs nothing), (c) a stride-0 per-edge stream would need a kernel
change, so instead (c) EK/EQ tables shrunk to ONE molecule's rows (every gather hits the same 21 atoms: L2-resident)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gotennet_amd import synthetic
from gotennet_amd._lib import call, ptr
from gotennet_amd.graph import distance
F = 256
wl, nmol = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("rmd17_aspirin", 128)
pos, batch, z = synthetic.make_batch(wl, nmol, seed=0)
ei, ed, ev = distance(pos.cuda(), batch.cuda(), 5.0, 32)
N, E = pos.shape[0], ei.shape[1]
src, dst = ei[0].to(torch.int32).contiguous(), ei[1].to(torch.int32).contiguous()
rowptr = torch.zeros(N + 1, dtype=torch.int32, device="cuda"); rowptr[1:] = torch.cumsum(torch.bincount(dst.long(), minlength=N), 0)
st = torch.cuda.current_stream().cuda_stream
r = lambda *s: torch.randn(*s, device="cuda")


def csc(s):
    colptr, perm, tgt = torch.empty(N + 1, dtype=torch.int32, device="cuda"), torch.empty(E, dtype=torch.int32, device="cuda"), torch.empty(E, dtype=torch.int32, device="cuda")
    work = torch.empty(N + E, dtype=torch.int32, device="cuda")
    call("gn_build_csc", ptr(s), ptr(dst), E, N, ptr(colptr), ptr(perm), ptr(tgt), ptr(work), st)
    return colptr, perm, tgt


def timeit(f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 50


for lmax in (2, 3, 4):
    D = (lmax + 1) ** 2 - 1
    EQ, EK, rl, w = r(N, D, F), r(N, D, F), r(E, D), torch.empty(E, F, device="cuda")
    gt, pre_t, gEQ, gEK, g_rl, g_pre = r(E, F), r(E, F), torch.empty(N, D, F, device="cuda"), torch.empty(N, D, F, device="cuda"), torch.empty(E, D, device="cuda"), torch.empty(E, F, device="cuda")
    for name, s in (("real sources", src), ("source = target", dst.clone())):
        colptr, perm, tgt = csc(s)
        t7 = timeit(lambda: call("gn_htr_edge", ptr(EQ), ptr(EK), ptr(rl), ptr(rowptr), ptr(s), N, F, lmax, 0, None, ptr(w), st))
        tb = timeit(lambda: call("gn_htr_backward", ptr(gt), ptr(pre_t), ptr(w), None, ptr(EQ), ptr(EK), ptr(rl), ptr(rowptr), ptr(s), ptr(tgt),
                                 ptr(colptr), ptr(perm), N, F, lmax, 0, ptr(gEQ), ptr(gEK), ptr(g_rl), ptr(g_pre), 0, st))
        b7 = 4 * N * 2 * D * F + E * (4 * (F + D) + 16)
        bb = 4 * E * (4 * F + 2 * D) + 16 * E + 4 * N * 4 * D * F
        print(f"{wl} b={nmol} lmax {lmax} {name:16s}: K7 {t7:7.1f} us ({b7 / t7 / 8e6:.3f})   HTR backward {tb:7.1f} us ({bb / tb / 8e6:.3f})", flush=True)
