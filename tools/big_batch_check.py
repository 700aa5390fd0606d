"""Batch independence at sizes far above the bench's (index-width check): the first 128 molecules of a B-molecule batch
must give the energies and forces of the 128-molecule batch, bit for bit.   python tools/big_batch_check.py 1024 2048 4096"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import gotennet_amd
from gotennet_amd import engine, synthetic
engine.GEMM_MODE = os.environ.get("GN_GEMM_MODE", "split")      # bit-exact batch independence holds for the row-wise arithmetics
from gotennet_amd.graph import distance
from gotennet_amd.outputs import Atomwise
from gotennet_amd.pipeline import EnergyForces
torch.manual_seed(0)
lmax = int(os.environ.get("LMAX", "2"))
net = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                            num_heads=8, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).cuda().eval()
head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").cuda().eval()
ef = EnergyForces(net, head)

def run(B):
    pos, batch, z = synthetic.make_batch("rmd17_aspirin", B, seed=0)
    pos, batch, z = pos.cuda(), batch.cuda(), z.cuda()
    ei, w, vec = distance(pos, batch, 5.0, 32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e, f = ef(z, ei, w, vec, batch, B)
    torch.cuda.synchronize()
    return e, f, ei.shape[1], time.perf_counter() - t0, torch.cuda.max_memory_allocated() / 2**30

e0, f0, E0, _, _ = run(128)
for B in [int(a) for a in sys.argv[1:]]:
    e, f, E, dt, gb = run(B)
    n = 128 * 21
    same = torch.equal(e[:128], e0) and torch.equal(f[:n], f0)
    tail_ok = bool(torch.isfinite(e).all() and torch.isfinite(f).all())
    net_force = float(f.reshape(B, 21, 3).sum(1).abs().max() / f.abs().max())
    print(f"B={B} N={B*21} E={E} E*(1+M)F={E * 1536 / 2**31:.2f} x 2^31  first-128 identical: {same}  finite: {tail_ok}  "
          f"max net force / max force {net_force:.1e}  {1e3*dt:.0f} ms  peak {gb:.1f} GiB", flush=True)
    assert same and tail_ok and net_force < 1e-3
