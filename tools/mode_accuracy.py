"""Worst error against the reference goldens per projection arithmetic (GPU box): max over the fixtures of the max-norm
relative error of h, X, energy, forces, and the same against the fixtures' fp64 truth."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gotennet_amd import engine
from gotennet_amd.pipeline import EnergyForces
from tests.golden_util import case_names, load_case, rel_err
from tests.test_hip_parity import _net_from_case
from tests.test_hip_forces import _head_from_case
names = [n for n in case_names() if "shuffled" not in n]
print(f"{len(names)} fixtures")
for mode in ("f32", "split", "f16x2"):
    engine.GEMM_MODE = mode
    worst = dict(h=0.0, X=0.0, e=0.0, f=0.0, h64=0.0, f64=0.0)
    for name in names:
        cfg, sd, head_sd, t = load_case(name)
        net, head = _net_from_case(cfg, sd), _head_from_case(cfg, head_sd)
        args = (t["z"].cuda(), t["edge_index"].cuda(), t["edge_diff"].cuda(), t["edge_vec"].cuda())
        h, X = net(*args)
        pairs = [("h", h, t["h"]), ("X", X, t["X"]), ("h64", h, t["h_f64"])]
        if cfg.get("aggr", "add") != "max":          # aggr="max" is a forward-only fixture (no input-gradient kernel)
            e, f = EnergyForces(net, head)(*args, t["batch"].cuda(), cfg["n_mol"])
            pairs += [("e", e, t["energy"]), ("f", f, t["forces"]), ("f64", f, t["forces_f64"])]
        torch.cuda.synchronize()
        for k, a, b in pairs:
            worst[k] = max(worst[k], rel_err(a.cpu(), b))
    print(f"{mode:6s} vs reference fp32: h {worst['h']:.1e}  X {worst['X']:.1e}  E {worst['e']:.1e}  F {worst['f']:.1e}   "
          f"vs fp64 truth: h {worst['h64']:.1e}  F {worst['f64']:.1e}")
ref = 0.0
for name in names:
    cfg, sd, head_sd, t = load_case(name)
    ref = max(ref, rel_err(t["forces"], t["forces_f64"]))
print(f"reference fp32 itself vs fp64 truth: F {ref:.1e}")
