"""Wall time of the radius-graph builder (Distance.forward equivalent) on the bench workload."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gotennet_amd import synthetic
from gotennet_amd.graph import distance
pos, batch, z = synthetic.make_batch("rmd17_aspirin", 128, seed=0)
pos, batch = pos.cuda(), batch.cuda()
for _ in range(3): ei, w, v = distance(pos, batch, 5.0, 32)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50): ei, w, v = distance(pos, batch, 5.0, 32)
torch.cuda.synchronize()
print("distance(): %.1f us per call, E=%d" % ((time.perf_counter() - t0) / 50 * 1e6, ei.shape[1]))
