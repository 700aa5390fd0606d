"""Where is K6 bound?  The same launch with (a) the real source indices, (b) every edge's source = its target (the
gathered rows then come from L1/L2-hot lines: no L2 -> CU gather traffic to speak of), (c) sources shuffled over the
whole batch (no L2 locality), (d) every edge reading the SAME t_filter row (row stride 0: the per-edge stream costs nothing --
what a perfect prefetch of that stream could reach)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gotennet_amd import synthetic
from gotennet_amd._lib import call, ptr
from gotennet_amd.graph import distance
N_MOL, F, H, lmax, M, D = 128, 256, 8, 2, 5, 8
pos, batch, z = synthetic.make_batch("rmd17_aspirin", N_MOL, seed=0)
ei, ed, ev = distance(pos.cuda(), batch.cuda(), 5.0, 32)
N, E = pos.shape[0], ei.shape[1]
src, dst = ei[0].to(torch.int32), ei[1].to(torch.int32)
rowptr = torch.zeros(N + 1, dtype=torch.int32, device="cuda"); rowptr[1:] = torch.cumsum(torch.bincount(dst.long(), minlength=N), 0)
r = lambda *s: torch.randn(*s, device="cuda")
x, v, tf, a, rl, cut = r(N, M * F), r(N, M * F), r(E, (1 + M) * F), r(E, H), r(E, D), r(E)
h, X, h2, X2 = r(N, F), r(N, D, F), torch.empty(N, F, device="cuda"), torch.empty(N, D, F, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(s, ldt=(1 + M) * F):
    f = lambda: call("gn_message_aggregate", ptr(x), ptr(v), M * F, tf.data_ptr() + 4 * F, ldt, ptr(a), ptr(rl), ptr(cut),
                     ptr(rowptr), ptr(s), ptr(h), ptr(X), ptr(h2), ptr(X2), N, F, H, lmax, 1, 1, st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 50
print("real sources      : %.1f us" % run(src))
print("source = target   : %.1f us" % run(dst.clone()))
print("shuffled sources  : %.1f us" % run(src[torch.randperm(E, device="cuda")].contiguous()))
print("one t_filter row  : %.1f us (real sources, row stride 0)" % run(src, 0))
