#!/usr/bin/env python
"""bench.py -- energy+force throughput of the GotenNet hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--lmax 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one energy+force evaluation (representation forward, Atomwise head,
hand-written backward, force scatter) of one synthetic batch per rank:
rMD17-aspirin-like molecules (21 atoms, uniform in a 4.6 A cube, 5 A radius graph
with self-loops; BASELINE.json configs[1]: batch 128, n_atom_basis 256,
n_interactions 6, reference yaml flags lmax 2 / sep_dir / sep_tensor / 8 heads).
Inputs (z, edge_index, edge_diff, edge_vec, batch) are resident in HBM before the
timed region.  N > 1: molecules are sharded by batch index (rank r owns molecules
[128 r, 128 (r+1))), weights replicated, ONE RCCL all-reduce per step on the
zero-padded energy vector; forces stay shard-local (weak scaling).

Prints one JSON line (rank 0).  `roofline` is measured live with HIP events around
every launch of the dominant kernel inside the timed region; `cpu_baseline` times the
CPU oracle (oracle/, a restatement pinned to the reference's golden vectors) on a
bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

JSON_FD = 1
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # v_mfma_f32_32x32x2_f32, exact fp32


class KernelTimer:
    def __init__(self, wanted=None):
        self.wanted, self.events = wanted, []

    @staticmethod
    def tag_of(name, args):
        if name in ("gn_gemm_ex", "gn_gemm_split"):
            return f"gn_gemm[{args[6]}x{args[7]}x{args[8]}]"
        if name == "gn_gemm_group":            # several independent problems in one launch
            return "gn_gemm[" + "+".join(f"{args[0][i].M}x{args[0][i].N}x{args[0][i].K}" for i in range(args[1])) + "]"
        return name

    def want(self, name, args):
        tag = self.tag_of(name, args)
        return tag if (self.wanted is None or tag in self.wanted) else None

    def summary(self):
        tot, cnt = {}, {}
        for tag, e0, e1 in self.events:
            tot[tag] = tot.get(tag, 0.0) + e0.elapsed_time(e1)
            cnt[tag] = cnt.get(tag, 0) + 1
        return tot, cnt


def _engine_mode():
    from gotennet_amd import engine
    return engine.GEMM_MODE


def family(tag):
    """Kernel family of a launch tag: every projection launch is the same MFMA kernel template."""
    return "gn_gemm" if tag.startswith("gn_gemm[") else tag


def algorithmic_bytes_message(N, E, F, M, D):
    """SURVEY.md 8(d) B_msg: every distinct input element read once, every output written once."""
    return 4 * N * (2 * F + 2 * M * F + D * F) + E * (4 * (F + M * F + D + 2) + 16) + 4 * N * (F + D * F)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="molecules per GPU")
    ap.add_argument("--lmax", type=int, default=2)
    ap.add_argument("--workload", default="rmd17_aspirin")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel table (stderr)")
    ap.add_argument("--no-lmax4", action="store_true", help="skip the short lmax=4 side measurement")
    ap.add_argument("--no-split", action="store_true", help="skip the short 3xbf16-split side measurement")
    ap.add_argument("--no-graph", action="store_true", help="skip the single-molecule hipGraph-replay side measurement")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even with one rank (path check)")
    a = ap.parse_args()

    # keep stdout clean for the ONE JSON line: RCCL prints a version banner to C stdout at exit
    global JSON_FD
    JSON_FD = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    res = measure(a, a.lmax, a.steps, a.warmup, rank, world, dev, dist)
    side = None
    if a.lmax != 4 and not a.no_lmax4:
        # the north-star's "L=4" target shape (lmax = 4), reported alongside (short run)
        side = measure(a, 4, max(3, a.steps // 4), 2, rank, world, dev, dist)
    split = None
    if not a.no_split and os.environ.get("GN_GEMM_MODE", "f32") == "f32":
        # opt-in projection mode (3 x bf16-split MFMA, fp32-class error), reported alongside
        from gotennet_amd import engine
        engine.GEMM_MODE = "split"
        try:
            split = measure(a, a.lmax, max(3, a.steps // 4), 2, rank, world, dev, dist)
        finally:
            engine.GEMM_MODE = "f32"
    lat = None
    if not a.no_graph and world == 1:
        lat = graph_latency(a, res["rep"], res["head"], dev)
    if rank == 0:
        out = res["out"]
        if lat is not None:
            out.setdefault("also", {})["single_molecule_latency"] = lat
        if split is not None:
            so = split["out"]
            out.setdefault("also", {})["split_bf16x3_projections"] = {
                "value": so["value"], "unit": so["unit"], "ms_per_step": so["ms_per_step"], "steps": so["steps"],
                "note": "GN_GEMM_MODE=split: every fp32 operand as hi+mid+lo bf16 planes, 6 bf16 MFMAs per product, "
                        "fp32 accumulate; same 1e-4 parity tests pass (error vs fp64 equals the exact-fp32 path)"}
        if side is not None:
            so = side["out"]
            out.setdefault("also", {})["lmax4"] = {k: so[k] for k in ("value", "unit", "ms_per_step", "steps",
                                                                       "roofline", "roofline_gather_scatter")}
            out["also"]["lmax4"]["config"] = so["config"]["workload"]
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(res["rep"], res["head"], a.workload, a.lmax)
        os.write(JSON_FD, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


def graph_latency(a, rep, head, dev, n_mol=1, iters=200):
    """One molecule of the workload (the MD use case): the eager fused step is launch-bound; pipeline.CapturedStep
    replays the same launches from ONE hipGraph (static topology).  Same model as the headline line."""
    from gotennet_amd import synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import CapturedStep, EnergyForces
    pos, batch, z = synthetic.make_batch(a.workload, n_mol, seed=0)
    pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    ef = EnergyForces(rep, head)

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / iters

    eager = timed(lambda: ef(z, ei, ed, ev, batch, n_mol))
    step = CapturedStep(ef, z, ei, batch, n_mol)
    e_g, f_g = step(pos)
    e_e, f_e = ef(z, ei, ed, ev, batch, n_mol)
    same = bool(torch.equal(e_g, e_e) and torch.equal(f_g, f_e))
    replay = timed(lambda: step(pos))
    return {"molecules": n_mol, "atoms": int(pos.shape[0]), "edges": int(ei.shape[1]),
            "eager_ms_per_step": round(eager, 3), "hipgraph_replay_ms_per_step": round(replay, 3),
            "steps_per_s_hipgraph": round(1e3 / replay, 1), "bit_identical_to_eager": same,
            "note": "static topology (fixed edge list, new positions every step): ~190 launches replayed as one hipGraph"}


def measure(a, lmax, steps, warmup, rank, world, dev, dist):
    import gotennet_amd
    from gotennet_amd import _lib, synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise, molecule_ptr
    from gotennet_amd.pipeline import EnergyForces

    F, L, R, H = 256, 6, 32, 8
    torch.manual_seed(0)                                   # identical replicated weights on every rank
    rep = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=L, n_rbf=R, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=H, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
    head = Atomwise(n_in=F, n_hidden=256, derivative="forces").to(dev).eval()
    step_fn = EnergyForces(rep, head)

    B = a.batch
    pos, batch, z = synthetic.make_batch(a.workload, B, seed=0, first_molecule=rank * B)
    pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    mol_ptr = molecule_ptr(batch, B)
    N, E = pos.shape[0], ei.shape[1]
    M, D = rep.config().M, rep.config().D
    from gotennet_amd.parallel import reduce_energies
    e_all = torch.zeros(B * world, dtype=torch.float32, device=dev)

    def step():
        e, f = step_fn(z, ei, ed, ev, batch, B, mol_ptr=mol_ptr)
        if dist is not None:                               # the one data-path collective (RCCL over xGMI)
            reduce_energies(e[:, 0], rank * B, B * world, out=e_all)
        return e, f

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed: warm-up + per-kernel breakdown to pick the dominant kernel ----------
    for _ in range(max(warmup, 1)):
        step()
    fence()
    kt = KernelTimer()
    _lib.TIMER = kt
    step()
    torch.cuda.synchronize()
    _lib.TIMER = None
    tot, cnt = kt.summary()
    fam = {}
    for tag, t in tot.items():
        fam[family(tag)] = fam.get(family(tag), 0.0) + t
    dominant = max(fam, key=fam.get)               # kernel family with the largest share of the step
    msg_tag = "gn_message_aggregate"
    if a.breakdown and rank == 0:
        ssum = sum(tot.values())
        print(f"# lmax={lmax}: per-kernel HIP-event breakdown of one step ({ssum:.3f} ms of events)", file=sys.stderr)
        for tag in sorted(tot, key=tot.get, reverse=True):
            print(f"  {tag:42s} {tot[tag]:8.3f} ms/step {cnt[tag]:3d} calls {1e3 * tot[tag] / cnt[tag]:8.1f} us/call "
                  f"{100 * tot[tag] / ssum:5.1f}%", file=sys.stderr)

    # ---- timed region: exactly K steps, events only around the dominant + message kernels
    # (the ~130 projection launches of a step are bracketed on the LAST timed step only, so that the
    # event records do not perturb `value`; the 6 message launches are bracketed on every step)
    dom_tags = {t for t in tot if family(t) == dominant}
    kt = KernelTimer(wanted={msg_tag} if len(dom_tags) > 8 else dom_tags | {msg_tag})
    _lib.TIMER = kt
    fence()
    t0 = time.perf_counter()
    for it in range(steps):
        if it == steps - 1:
            kt.wanted = dom_tags | {msg_tag}
        e, f = step()
    fence()
    dt = time.perf_counter() - t0
    _lib.TIMER = None
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(e).all() and torch.isfinite(f).all()

    tot, cnt = kt.summary()

    dom_steps = 1 if len(dom_tags) > 8 else steps

    def roof_gemm_family():
        """All projection launches of the timed region: algorithmic flops / summed launch time."""
        flops = t_ms = 0.0
        n = 0
        big = None
        for tag in tot:
            if family(tag) != "gn_gemm":
                continue
            fl = sum(2.0 * m_ * n_ * k_ for m_, n_, k_ in
                     (tuple(int(v) for v in part.split("x")) for part in tag[8:-1].split("+")))
            flops += fl * cnt[tag]
            t_ms += tot[tag]
            n += cnt[tag]
            us = 1e3 * tot[tag] / cnt[tag]
            if big is None or tot[tag] > big[1]:
                big = (tag, tot[tag], us, fl / (us * 1e-6) / 1e12)
        ach = flops / (t_ms * 1e-3) / 1e12
        from gotennet_amd import engine as _eng
        exact = _eng.GEMM_MODE == "f32"
        return dict(kernel="gn::gemm_f32_mfma (all projection launches, exact fp32 MFMA)" if exact else
                    "gn::gemm_bf16x3_mfma (all projection launches, 3xbf16-split MFMA)",
                    bound="mfma", achieved=round(ach, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                    frac=round(ach / MFMA_F32_PEAK_TF, 4), traffic=_pmc_traffic("gn_gemm_family_avg", lmax),
                    us_per_launch=round(1e3 * t_ms / n, 2), launches_per_step=n // dom_steps,
                    algorithmic_flops_per_step=flops / dom_steps,
                    largest_launch=dict(shape_MxNxK=big[0][8:-1], us=round(big[2], 2), tflops=round(big[3], 2),
                                        frac=round(big[3] / MFMA_F32_PEAK_TF, 4)))

    def roof_message():
        us = 1e3 * tot[msg_tag] / cnt[msg_tag]
        nbytes = algorithmic_bytes_message(N, E, F, M, D)
        ach = nbytes / (us * 1e-6) / 1e9
        return dict(kernel=msg_tag, bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 4), traffic=_pmc_traffic(msg_tag, lmax), us_per_launch=round(us, 2),
                    launches_per_step=cnt[msg_tag] // steps, algorithmic_bytes_per_launch=nbytes)

    def roof_other(name):
        t_ms = sum(tot[t] for t in tot if family(t) == name)
        n = sum(cnt[t] for t in tot if family(t) == name)
        return dict(kernel=name, bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None,
                    us_per_launch=round(1e3 * t_ms / n, 2), launches_per_step=n // dom_steps)

    out = None
    if rank == 0:
        out = {
            "metric": "molecules/sec (energy+force forward), rMD17 aspirin batch=128, 1/2/4/8 MI355X",
            "value": round(B * world * steps / dt, 1), "unit": "molecules/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if _engine_mode() == "f32" else "f32 (3xbf16-split MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{a.workload} batch={B}/GPU (N={N} atoms, E={E} edges incl. self-loops), "
                                   f"n_atom_basis={F}, n_interactions={L}, lmax={lmax}, n_rbf={R}, heads={H}, "
                                   "sep_dir/sep_tensor, energy+forces",
                       "global_batch": B * world, "parallelism": f"dp{world} (molecule shards, 1 all-reduce)"},
            "roofline": roof_gemm_family() if dominant == "gn_gemm" else
            (roof_message() if dominant == msg_tag else roof_other(dominant)),
            "roofline_gather_scatter": roof_message(),
        }
    return {"out": out, "rep": rep, "head": head}


def _pmc_traffic(tag, lmax):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), if present."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return d.get(f"lmax{lmax}", {}).get(tag)
    except Exception:
        return None


def cpu_baseline(rep, head, workload, lmax, n_mol=8, reps=2):
    """The CPU oracle (checker) timed on this box's host cores: energy+forces via autograd
    on a bounded sample (n_mol molecules of the same workload, same hyper-parameters)."""
    from gotennet_amd import synthetic
    from oracle import gotennet_oracle as orc
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    sd = {k: v.detach().cpu() for k, v in rep.state_dict().items()}
    hsd = {k: v.detach().cpu() for k, v in head.state_dict().items()}
    c = rep.config()
    cfg = orc.default_config(n_atom_basis=c.F, n_interactions=c.L, n_rbf=c.R, num_heads=c.H, scale_edge=c.scale_edge,
                             lmax=lmax, sep_dir=c.sep_dir, sep_tensor=c.sep_tensor, cutoff=c.cutoff)
    pos, batch, z = synthetic.make_batch(workload, n_mol, seed=0)
    orc.energy_and_forces(sd, cfg, hsd, z, pos, batch, n_mol)           # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.energy_and_forces(sd, cfg, hsd, z, pos, batch, n_mol)
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(n_mol / dt, 2), "unit": "molecules/s", "cores": threads, "kind": "port",
            "sample": f"{n_mol} molecules of {workload} (same model), energy+forces by torch autograd on the CPU oracle, "
                      f"{reps} runs after 1 warm-up, {threads} threads"}


if __name__ == "__main__":
    main()
