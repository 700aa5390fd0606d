"""One 128-molecule batch as 1 / 2 / 3 / 4 sub-batches on as many InFlight lanes (all sub-batches of a batch are issued, then
the next batch): ms per FULL batch.  Is a split batch faster than the whole one (the launches of one step are a dependent
chain; sub-batches on separate streams overlap matrix and memory kernels) -- or do small sub-batches fill the chip too badly?"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import gotennet_amd
from gotennet_amd import synthetic
from gotennet_amd.graph import distance
from gotennet_amd.outputs import Atomwise
from gotennet_amd.pipeline import InFlight
dev = torch.device("cuda")
lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 128
torch.manual_seed(0)
rep = gotennet_amd.GotenNet(n_atom_basis=256, n_interactions=6, n_rbf=32, cutoff_fn=gotennet_amd.CosineCutoff(5.0), num_heads=8,
                            scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
head = Atomwise(n_in=256, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
for parts in (1, 2, 3, 4, 1, 2):
    sub = B // parts
    data = []
    for q in range(parts):
        nb = sub if q < parts - 1 else B - sub * (parts - 1)
        pos, batch, z = synthetic.make_batch("rmd17_aspirin", nb, seed=0, first_molecule=q * sub)
        pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)
        data.append((z, *distance(pos, batch, 5.0, 32), batch, nb))
    fl = InFlight(rep, head, lanes=parts, check_edges=False, cache_topology=False)
    def full():
        for z, ei, ed, ev, batch, nb in data:
            fl(z, ei, ed, ev, batch, nb)
        fl.wait()
    for _ in range(4): full()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 15
    for _ in range(n):
        full()
        torch.cuda.current_stream().synchronize()          # one batch at a time: the next batch starts when this one is done
    dt = (time.perf_counter() - t0) / n
    print(f"lmax {lmax}: {parts} sub-batch(es) of {sub}: {1e3 * dt:.3f} ms per 128-molecule batch ({B / dt:.0f} molecules/s)", flush=True)
