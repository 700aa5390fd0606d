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
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md)


class KernelTimer:
    def __init__(self, wanted=None):
        self.wanted, self.events = wanted, []

    @staticmethod
    def tag_of(name, args):
        if name in ("gn_gemm_ex", "gn_gemm_split", "gn_gemm_f16x2"):
            return f"gn_gemm[{args[6]}x{args[7]}x{args[8]}]"
        if name in ("gn_gemm_group", "gn_gemm_group_split", "gn_gemm_group_f16x2"):   # several independent problems in one launch
            # (a trailing "g" marks a problem with the gated-residual epilogue C = res + act(.) * gate: the HTR edge update)
            return "gn_gemm[" + "+".join(f"{args[0][i].M}x{args[0][i].N}x{args[0][i].K}"
                                         + ("g" if (args[0][i].gate and not args[0][i].gate_mode and args[0][i].res) else "")
                                         for i in range(args[1])) + "]"
        return name

    #: X_in argument of the two message entries: None = the zero-X_in launch of the first interaction
    X_IN_ARG = {"gn_message_aggregate": 11, "gn_message_backward": 8}

    def want(self, name, args):
        tag = self.tag_of(name, args)
        if self.wanted is not None and tag not in self.wanted:
            return None
        k = self.X_IN_ARG.get(name)
        return tag + "|first" if (k is not None and args[k] is None) else tag

    def summary(self, split_first=False):
        """Total ms and launch count per tag; ``split_first`` keeps the first interaction's launches apart."""
        tot, cnt = {}, {}
        for tag, e0, e1 in self.events:
            if not split_first:
                tag = tag.split("|")[0]
            tot[tag] = tot.get(tag, 0.0) + e0.elapsed_time(e1)
            cnt[tag] = cnt.get(tag, 0) + 1
        return tot, cnt


def _engine_mode():
    from gotennet_amd import engine
    return engine.GEMM_MODE


def family(tag):
    """Kernel family of a launch tag: every projection launch is the same MFMA kernel template."""
    return "gn_gemm" if tag.startswith("gn_gemm[") else tag


def algorithmic_bytes_message(N, E, F, M, D, first_nd=None):
    """SURVEY.md 8(d) B_msg: every distinct input element read once, every output written once.
    ``first_nd`` (the number of direction-gate blocks): the launch of the FIRST interaction, whose X_in is identically
    zero -- no tensor-gate blocks of t_filter / x / v and no X_in table (gotennet_amd.engine.zero_X_in)."""
    if first_nd is not None:
        Mv = 1 + first_nd
        return 4 * N * (2 * F + 2 * Mv * F) + E * (4 * (F + Mv * F + D + 2) + 16) + 4 * N * (F + D * F)
    return 4 * N * (2 * F + 2 * M * F + D * F) + E * (4 * (F + M * F + D + 2) + 16) + 4 * N * (F + D * F)


#: the driver keeps an 8 KB tail of stdout and parses the last line: the line printed to stdout stays far below that
#: (round 3's 21 KB line was not parsed: BENCH_r03.parsed = null).  The full record goes to a side file and to stderr.
LINE_BUDGET = 6000

_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "launches_per_step",
              "algorithmic_bytes_per_launch", "algorithmic_tflops", "executed_terms_per_product", "mfma_busy")


def _roof_compact(r, name_len=60):
    """One roofline record, one level deep: the contract keys + what prices them; no notes, no nested copies."""
    if not r:
        return None
    o = {k: r[k] for k in _ROOF_KEYS if k in r}
    o["kernel"] = str(o.get("kernel", ""))[:name_len].split(" (")[0]
    big = r.get("largest_launch")
    if big:
        o["largest_launch"] = {"shape": big.get("shape_MxNxK"), "us": big.get("us"), "frac": big.get("frac")}
    for k in ("general_launches", "first_launch"):
        if k in r:
            o[k] = {"us": r[k].get("us_per_launch"), "frac": r[k].get("frac")}
    return o


def _frac(r, sub=None):
    if not r:
        return None
    if sub and sub in r:
        return r[sub].get("frac")
    return r.get("frac")


def _side_compact(so):
    """A side workload: value, time and the four family fractions (gather/scatter stage, K7, message backward,
    projections)."""
    g = so.get("roofline_gather_scatter")
    o = {"value": so.get("value"), "ms_per_step": so.get("ms_per_step"), "steps": so.get("steps"),
         "gather_frac": _frac(g), "gather_frac_general": _frac(g, "general_launches"),
         "htr_frac": _frac(so.get("roofline_htr_edge")), "msg_bwd_frac": _frac(so.get("roofline_message_backward")),
         "htr_bwd_frac": _frac(so.get("roofline_htr_backward")),
         "gemm_frac": _frac(so.get("roofline")), "gemm_alg_tflops": (so.get("roofline") or {}).get("algorithmic_tflops")}
    return {k: v for k, v in o.items() if v is not None}


def compact_line(full):
    """The ONE JSON line of the bench contract, bounded (< LINE_BUDGET bytes): headline keys, `roofline` (dominant kernel
    family), the three stage records, `cpu_baseline`, and a compact `also` (one small dict per side measurement).
    ``full`` is the complete record (what round 3 printed); it is written to a side file by ``emit``."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config", "n_ranks_seen", "energy_vector_len",
                                "energy_checksum", "rank_ms_per_step", "launch_mode", "batches_in_flight",
                                "value_one_at_a_time", "ms_per_batch_one_at_a_time", "roofline_target",
                                "lanes_consistent", "lanes_fallback")
           if k in full and not (k == "rank_ms_per_step" and full[k] is None)}
    out["roofline"] = _roof_compact(full.get("roofline"))
    if out["roofline"] is not None:
        live = str((full.get("roofline") or {}).get("traffic_source", "")).startswith("measured in this run")
        out["roofline"]["traffic_source"] = ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE sub-runs of this bench (2 x FETCH + WRITE)"
                                             if live else "committed profiles/pmc_traffic.json (rocprofv3 --pmc; FETCH_SIZE x2 + WRITE_SIZE)")
    for k in ("roofline_gather_scatter", "roofline_htr_edge", "roofline_message_backward", "roofline_htr_backward",
              "roofline_gated_gemm"):
        if full.get(k):
            out[k] = _roof_compact(full[k])
            if k in ("roofline_htr_backward", "roofline_gated_gemm"):      # (the byte formula travels with the two new records)
                out[k]["bytes"] = str(full[k].get("note", ""))[:96]
    also_f, also = full.get("also") or {}, {}
    for k, v in also_f.items():
        if k == "other_projection_modes":
            also[k] = {m: {"value": o.get("value"), "ms_per_step": o.get("ms_per_step"), "gemm_frac": _frac(o.get("roofline")),
                           "gemm_alg_tflops": (o.get("roofline") or {}).get("algorithmic_tflops")} for m, o in v.items()}
        elif k == "forward_only":
            also[k] = {"value": v.get("value"), "ms_per_step": v.get("ms_per_step"), "steps": v.get("steps"),
                       "cpu_value": (v.get("cpu_baseline") or {}).get("value")}
        elif k == "batches_in_flight":
            also[k] = {n: {"value": o.get("value"), "ms_per_batch": o.get("ms_per_batch")} for n, o in v.items()}
        elif k == "static_topology":
            also[k] = {"value": v.get("value"), "ms_per_step": v.get("ms_per_step"), "steps": v.get("steps")}
        elif k in ("single_molecule_latency", "hipgraph_replay_full_batch"):
            also[k] = {kk: v.get(kk) for kk in ("molecules", "eager_ms_per_step", "hipgraph_replay_ms_per_step",
                                                "bit_identical_to_eager")}
        elif k == "lmax4" and full.get("roofline_target"):
            continue                                 # (the top-level roofline_target carries the same record)
        elif isinstance(v, dict) and "value" in v:
            also[k] = _side_compact(v)
            if isinstance(v.get("config"), str):
                also[k]["config"] = v["config"][:80]
    if also:
        out["also"] = also
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "cpu_model", "host_cores_physical",
                                                   "thread_sweep_8_molecules", "largest_batch") if k in cb}
        out["cpu_baseline"]["sample"] = str(cb.get("sample_short") or cb.get("sample", ""))[:160]
    if full.get("full_record"):
        out["full_record"] = full["full_record"]
    line = json.dumps(out)
    if len(line) >= LINE_BUDGET:                     # never over budget: shed the optional part, keep the contract keys
        out.pop("also", None)
        out["also_dropped"] = "line over budget; see full_record"
        line = json.dumps(out)
    return line


def emit(full, path=None):
    """Full record -> side file (+ stderr); compact line -> the saved stdout descriptor (the only thing on stdout)."""
    if path is None:
        path = os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(full, fh)
        full["full_record"] = os.path.relpath(path, ROOT)
    except OSError:
        pass
    print(f"# full record: {full.get('full_record', '(not written)')}", file=sys.stderr)
    os.write(JSON_FD, (compact_line(full) + "\n").encode())


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
    ap.add_argument("--no-split", action="store_true", help="skip the short side measurements in the other projection arithmetics")
    ap.add_argument("--no-graph", action="store_true", help="skip the single-molecule hipGraph-replay side measurement")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the energy-only (no force backward) side measurement")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even with one rank (path check)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the short C3 / C5 side measurements")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from the committed PMC passes instead of two rocprofv3 sub-runs of this script")
    ap.add_argument("--static-topology", action="store_true",
                    help="time the step on a cached topology (CSR / CSC / molecule offsets built once): an MD loop on a fixed neighbour list")
    ap.add_argument("--no-static", action="store_true", help="skip the static-topology / batches-in-flight side measurements")
    ap.add_argument("--lanes", type=int, default=3,
                    help="batches in flight: steps are fed round-robin to this many EnergyForces lanes on their own HIP streams "
                         "(pipeline.InFlight); 1 = one step at a time")
    ap.add_argument("--replay", action="store_true", help="hipGraph replay of the static-topology step (EnergyForces(replay=True)) instead of eager launches")
    ap.add_argument("--full-json", default=None,
                    help="where the FULL record goes (default: gpurun_out/bench_full.json next to this file, when that "
                         "directory can be created); stdout carries the compact line only")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="CPU-only check of the launch / rendezvous / shard / all-reduce / JSON path (gloo backend, the step "
                         "replaced by a per-molecule checksum of the synthetic inputs); no kernel runs, no throughput claim")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)                      # plain `python bench.py --gpus N`: spawn one rank per GPU
    worker(a)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawned(local_rank, a, port):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(a.gpus),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this driver)
    worker(a)


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: one process per GPU via torch.multiprocessing.spawn
    (the same ranks `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` would start)."""
    import torch.multiprocessing as mp
    if not a.selftest_dist and torch.cuda.device_count() < a.gpus:
        raise SystemExit(f"--gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
    mp.spawn(_spawned, args=(a, _free_port()), nprocs=a.gpus, join=True)


def selftest_dist(a, rank, world):
    """The distributed skeleton of the bench without a GPU: rendezvous (gloo), molecule shards, the one all-reduce of
    the zero-padded per-molecule vector, max-over-ranks timing, rank-0 JSON.  The per-molecule value is a checksum of
    the synthetic inputs (sum of z * |pos|^2) -- input synthesis, not path arithmetic."""
    import torch.distributed as dist
    from gotennet_amd import synthetic
    from gotennet_amd.parallel import reduce_energies
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = a.batch
    pos, batch, z = synthetic.make_batch(a.workload, B, seed=0, first_molecule=rank * B)
    val = torch.zeros(B, dtype=torch.float64).index_add_(0, batch, z.double() * (pos.double() ** 2).sum(1)).float()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        e_all = reduce_energies(val, rank * B, B * world)
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ok = bool(torch.equal(e_all[rank * B:(rank + 1) * B], val)) and int((e_all != 0).sum()) == B * world
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        out = {"selftest": "dist", "backend": "gloo", "n_gpus": 0, "n_ranks_seen": dist.get_world_size(),
               "global_batch": B * world, "steps": a.steps, "energy_vector_len": int(e_all.numel()),
               "energy_checksum": float(e_all.double().sum()), "shards_consistent": bool(flag.item() == 1.0),
               "ms_per_step": round(1e3 * float(tmax.item()) / max(a.steps, 1), 4)}
        os.write(JSON_FD, (json.dumps(out) + "\n").encode())
    dist.destroy_process_group()


def worker(a):
    # keep stdout clean for the ONE JSON line: RCCL prints a version banner to C stdout at exit
    global JSON_FD
    JSON_FD = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if a.selftest_dist:
        return selftest_dist(a, rank, world)
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

    from gotennet_amd import engine
    if world == 1 and rank == 0 and not a.no_live_traffic and not a.force_dist:
        # BEFORE this process touches the GPU's counters itself: the sub-runs profile the same headline workload
        lt = live_traffic(a)
        if lt:
            LIVE_TRAFFIC.update(lt, _key=(a.workload, a.batch, a.lmax))
    res = measure(a, a.workload, a.batch, a.lmax, a.steps, a.warmup, rank, world, dev, dist)
    one_at_a_time = None
    if world > 1 and res["lanes"] > 1:
        # N > 1: the same K steps one at a time as well (both figures in the record).  Should the laned run have lost a
        # shard or a rank (verdict shared by all ranks: an all-reduced flag), the one-at-a-time run IS the headline.
        import copy
        a1 = copy.copy(a)
        a1.lanes = 1
        res1 = measure(a1, a.workload, a.batch, a.lmax, a.steps, a.warmup, rank, world, dev, dist)
        if rank == 0:
            one_at_a_time = {"value": res1["out"]["value"], "ms_per_batch": res1["out"]["ms_per_step"]}
        if not res["lanes_ok"] or res["n_ranks_seen"] != world:
            if rank == 0:
                res1["out"]["lanes_fallback"] = (f"{res['lanes']} batches in flight failed the shard check "
                                                 f"(value {res['out']['value']}): headline = one step at a time")
            res = res1
    sides = world == 1                              # side measurements only on the single-GPU line
    side = lat = lat_batch = None
    wl = {}
    if sides and a.lmax != 4 and not a.no_lmax4 and a.workload == "rmd17_aspirin":
        # the north-star's "L=4" target shape (lmax = 4): the gather/scatter target is quoted on it
        side = measure(a, a.workload, a.batch, 4, max(20, a.steps), 3, rank, world, dev, dist)
    others = {}
    if sides and not a.no_split:
        # the other projection arithmetics (exact fp32 MFMA, 3 x bf16-split, 2 x fp16-split), reported alongside
        default_mode = engine.GEMM_MODE
        for mode in ("f32", "split", "f16x2"):
            if mode == default_mode:
                continue
            engine.GEMM_MODE = mode
            try:
                others[mode] = measure(a, a.workload, a.batch, a.lmax, max(5, a.steps // 2), 2, rank, world, dev, dist)
            finally:
                engine.GEMM_MODE = default_mode
    if sides and not a.no_workloads and a.workload == "rmd17_aspirin":
        # BASELINE configs[2] and configs[4] on the same model family (single GPU)
        wl["md22_ac_ala3_b64"] = measure(a, "md22_ac_ala3", 64, 2, 10, 2, rank, world, dev, dist)
        wl["md22_nanotube_b8_lmax3"] = measure(a, "md22_nanotube", 8, 3, 10, 2, rank, world, dev, dist)
    static = None
    if sides and not a.static_topology and not a.replay and not a.no_static:
        import copy
        a_st = copy.copy(a)
        a_st.static_topology, a_st.lanes = True, 1
        static = measure(a_st, a.workload, a.batch, a.lmax, max(10, a.steps // 2), 2, rank, world, dev, dist)
    inflight = None
    if sides and not a.no_static:                   # the same fresh-topology step at 1 / 2 / 3 batches in flight
        inflight = {str(n): in_flight(a, res["rep"], res["head"], dev, a.lmax, lanes=n) for n in (1, 2, 3)}
    elif sides:                                     # the one-at-a-time (latency) figure is always in the record
        inflight = {"1": in_flight(a, res["rep"], res["head"], dev, a.lmax, lanes=1)}
    if inflight is not None and rank == 0:
        one_at_a_time = {"value": inflight["1"]["value"], "ms_per_batch": inflight["1"]["ms_per_batch"]}
    fwd = None
    if sides and not a.no_forward_only:
        fwd = forward_only(a, res["rep"], res["head"], dev)
    if sides and not a.no_graph:
        lat = graph_latency(a, res["rep"], res["head"], dev)
        lat_batch = graph_latency(a, res["rep"], res["head"], dev, n_mol=a.batch, iters=20)
    if rank == 0:
        out = res["out"]
        also = out.setdefault("also", {})
        sub = lambda so: {**{k: so[k] for k in ("value", "unit", "ms_per_step", "steps", "roofline",
                                                 "roofline_gather_scatter", "roofline_htr_edge", "roofline_message_backward",
                                                 "roofline_htr_backward")},
                          "config": so["config"]["workload"]}
        if static is not None:
            so = static["out"]
            also["static_topology"] = {"value": so["value"], "ms_per_step": so["ms_per_step"], "steps": so["steps"],
                                       "note": "the same step with the CSR / CSC index arrays and molecule offsets cached across steps"}
        if inflight is not None:
            also["batches_in_flight"] = inflight
        if lat is not None:
            also["single_molecule_latency"] = lat
        if lat_batch is not None:                  # the headline batch as ONE hipGraph replay per step (fixed edge list)
            also["hipgraph_replay_full_batch"] = lat_batch
        if fwd is not None:
            if not a.no_cpu_baseline:
                fwd["cpu_baseline"] = cpu_baseline(res["rep"], res["head"], a.workload, a.lmax, forces=False)
            also["forward_only"] = fwd
        for mode, other in others.items():
            so = other["out"]
            also.setdefault("other_projection_modes", {})[mode] = {
                "dtype": so["dtype"], "value": so["value"], "unit": so["unit"], "ms_per_step": so["ms_per_step"],
                "steps": so["steps"], "roofline": so["roofline"]}
        if one_at_a_time is not None:               # top-level: the latency figure next to the in-flight throughput
            out["value_one_at_a_time"] = one_at_a_time["value"]
            out["ms_per_batch_one_at_a_time"] = one_at_a_time["ms_per_batch"]
        if side is not None:
            also["lmax4"] = sub(side["out"])
            # top-level copy of the north-star TARGET shape's record (n_atom_basis=256, lmax=4): >= 0.40 of the HBM roofline
            # on the edge gather / scatter is the stated target; the headline line is the reference's default lmax=2
            out["roofline_target"] = {"config": f"{a.workload} batch={a.batch}/GPU, n_atom_basis=256, n_interactions=6, lmax=4, energy+forces", **{
                k: v for k, v in _side_compact(side["out"]).items() if k in
                ("gather_frac", "gather_frac_general", "htr_frac", "msg_bwd_frac", "htr_bwd_frac", "ms_per_step", "value")}}
        for k, v in wl.items():
            also[k] = sub(v["out"])
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(res["rep"], res["head"], a.workload, a.lmax)
        emit(out, a.full_json)
    if dist is not None:
        dist.destroy_process_group()


def in_flight(a, rep, head, dev, lmax, steps=20, lanes=2):
    """Throughput with `lanes` batches in flight (`pipeline.InFlight`: EnergyForces lanes on their own HIP streams, fed
    round-robin; eager launches, fresh topology every call like the headline), a different batch per lane.  lanes = 1 is the
    one-step-at-a-time figure of earlier rounds."""
    from gotennet_amd import synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import InFlight
    B = a.batch
    data = []
    for q in range(lanes):                                   # a different batch per lane
        pos, batch, z = synthetic.make_batch(a.workload, B, seed=0, first_molecule=q * B)
        pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)
        data.append((z, *distance(pos, batch, 5.0, 32), batch))
    fl = InFlight(rep, head, lanes=lanes, check_edges=False, cache_topology=False)
    for it in range(2 * lanes):
        z, ei, ed, ev, batch = data[it % lanes]
        fl(z, ei, ed, ev, batch, B)
    fl.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        z, ei, ed, ev, batch = data[it % lanes]
        e, f = fl(z, ei, ed, ev, batch, B)
    fl.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(e).all() and torch.isfinite(f).all()
    return {"metric": f"molecules/sec (energy+force forward), {lanes} batch(es) in flight", "lanes": lanes,
            "value": round(B * steps / dt, 1), "unit": "molecules/s", "ms_per_batch": round(1e3 * dt / steps, 3), "steps": steps,
            "note": "pipeline.InFlight: the launches of one step are a dependent chain; a second step on a second stream puts its "
                    "memory-bound kernels beside the first step's power-capped matrix kernels.  Batch latency is not improved"}


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
            "note": "static topology (fixed edge list, new positions every step): the step's launches replayed as one hipGraph"}


def forward_only(a, rep, head, dev, steps=20):
    """The path's own API, forward only: representation forward + Atomwise energy, no force backward
    (`EnergyForces(..., forces=False)`: ping-pong work buffers, nothing saved).  Same workload and model as the headline,
    fresh topology every step like the headline."""
    from gotennet_amd import synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.pipeline import EnergyForces
    B = a.batch
    pos, batch, z = synthetic.make_batch(a.workload, B, seed=0)
    pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    ef = EnergyForces(rep, head, check_edges=False, cache_topology=False)
    for _ in range(3):
        e, _ = ef(z, ei, ed, ev, batch, B, forces=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e, _ = ef(z, ei, ed, ev, batch, B, forces=False)
    torch.cuda.synchronize()
    dt0 = time.perf_counter() - t0
    assert torch.isfinite(e).all()
    return {"metric": "molecules/sec (energy only: representation forward + Atomwise head, no forces)",
            "value": round(B * steps / dt0, 1), "unit": "molecules/s", "ms_per_step": round(1e3 * dt0 / steps, 3),
            "steps": steps, "config": f"{a.workload} batch={B}, same model as the headline line"}


#: launches of the GATA message stage (gotennet.py:452-559, 613-640): scores + segment softmax + message + aggregate.
#: B_msg is the stage's algorithmic traffic, so the stage's launches are timed TOGETHER.
MSG_STAGE = ("gn_attn_softmax", "gn_message_aggregate")
HTR_TAG = "gn_htr_edge"
MSGB_TAG = "gn_message_backward"


def algorithmic_bytes_message_backward(N, E, F, M, D, H):
    """Compulsory bytes of the message backward of one layer (the largest non-GEMM launch family of the force path):
    every distinct element read once / written once.  Reads: eproj [E,(1+M)F] (t_attn pre-activation + t_filter),
    a [E,H], rl [E,D], cut [E], four int32 index arrays, the node tables x, v [N,MF], q | k [N,2F], X_in, g_X1 [N,D,F],
    g_h1 [N,F].  Writes: g_eproj [E,(1+M)F], g_s [E,H], g_rl [E,D], g_cut [E], g_x, g_v [N,MF], g_q | g_k [N,2F],
    g_X [N,D,F]."""
    edge = 4 * E * (2 * (1 + M) * F + 2 * H + 2 * D + 2) + 16 * E
    node = 4 * N * (2 * M * F + 2 * F + 2 * D * F + F) + 4 * N * (2 * M * F + 2 * F + D * F)
    return edge + node


def algorithmic_bytes_message_backward_first(N, E, F, ND, D, H):
    """The same for the first interaction (X_in identically zero): no tensor-gate blocks in eproj / g_eproj / x / v / g_x /
    g_v, no X_in table, no g_X rows."""
    Mv = 1 + ND
    edge = 4 * E * (2 * (1 + Mv) * F + 2 * H + 2 * D + 2) + 16 * E
    node = 4 * N * (2 * Mv * F + 2 * F + D * F + F) + 4 * N * (2 * Mv * F + 2 * F)
    return edge + node


def measure(a, workload, B, lmax, steps, warmup, rank, world, dev, dist):
    import gotennet_amd
    from gotennet_amd import _lib, synthetic
    from gotennet_amd.graph import distance
    from gotennet_amd.outputs import Atomwise, molecule_ptr
    from gotennet_amd.pipeline import EnergyForces

    F, L, R, H = 256, 6, 32, 8
    torch.manual_seed(0)                                   # identical replicated weights on every rank
    rep = gotennet_amd.GotenNet(n_atom_basis=F, n_interactions=L, n_rbf=R, cutoff_fn=gotennet_amd.CosineCutoff(5.0),
                                num_heads=H, scale_edge=False, lmax=lmax, sep_dir=True, sep_tensor=True).to(dev).eval()
    head = Atomwise(n_in=F, n_hidden=256, derivative="forces", activation="silu").to(dev).eval()
    # radius-graph order is target-major by construction (no order check).  `--replay`: the product's static-topology
    # replay (pipeline.EnergyForces(replay=True): after two steps on one edge list the eager step is recorded into ONE
    # hipGraph and replayed on the step's fresh inputs; same launches, bit-identical).  Measured on the C2 batch: 7.90 vs
    # 7.84 ms eager -- the host runs 2x ahead of the GPU here, so the headline stays the eager path.
    # The HEADLINE pays what a fresh batch costs: no topology cache (the int64 -> CSR conversion, the by-source view, the
    # out-degree count run every step) and the molecule offsets are computed inside the step; `also.static_topology` is the
    # same step on a cached topology (an MD loop on a fixed neighbour list).
    fresh = not (a.static_topology or a.replay)
    step_fn = EnergyForces(rep, head, check_edges=False, replay=a.replay, cache_topology=not fresh)

    pos, batch, z = synthetic.make_batch(workload, B, seed=0, first_molecule=rank * B)
    pos, batch, z = pos.to(dev), batch.to(dev), z.to(dev)
    ei, ed, ev = distance(pos, batch, 5.0, 32)
    mol_ptr = molecule_ptr(batch, B)
    N, E = pos.shape[0], ei.shape[1]
    M, D = rep.config().M, rep.config().D
    from gotennet_amd.parallel import reduce_energies
    e_all = torch.zeros(B * world, dtype=torch.float32, device=dev)

    def step():
        e, f = step_fn(z, ei, ed, ev, batch, B, mol_ptr=None if fresh else mol_ptr)
        if dist is not None:                               # the one data-path collective (RCCL over xGMI)
            reduce_energies(e[:, 0], rank * B, B * world, out=e_all)
        return e, f

    # Batches in flight (--lanes > 1): consecutive steps go round-robin to `lanes` EnergyForces objects on their own HIP
    # streams -- the launches of ONE step are a dependent chain, so a second / third step's memory-bound kernels run beside
    # the first one's power-capped matrix kernels.  Every step is still one batch of B molecules through the
    # whole path incl. its all-reduce; the event-bracketed steps at the end of the timed region run ALONE (lanes drained).
    lanes = 1 if (a.replay or a.lanes < 2) else a.lanes
    fl = reducer = None
    if lanes > 1:
        from gotennet_amd.pipeline import InFlight
        fl = InFlight(rep, head, lanes=lanes, check_edges=False, cache_topology=not fresh)
        if dist is not None:
            # every lane's all-reduce leaves from ONE communication stream in submission order (parallel.OrderedReducer):
            # ranks whose lanes drift against each other still issue the same sequence of collectives
            from gotennet_amd.parallel import OrderedReducer
            reducer = OrderedReducer(B * world, rank * B, dev, slots=lanes)

    def step_lane():
        then = (lambda e_, f_: reducer.submit(e_[:, 0])) if reducer is not None else None
        return fl(z, ei, ed, ev, batch, B, mol_ptr=None if fresh else mol_ptr, _then=then)

    def fence():
        if fl is not None:
            fl.wait()
        if reducer is not None:
            reducer.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed: warm-up + per-kernel breakdown to pick the dominant kernel ----------
    for _ in range(max(warmup, 1)):
        step()
    if fl is not None:
        for _ in range(max(warmup, 1) * lanes):
            step_lane()
    fence()
    replay = step_fn.replay
    kt = KernelTimer()
    _lib.TIMER = kt
    step_fn.replay = False                                 # per-launch events need the eager launches
    step()
    torch.cuda.synchronize()
    step_fn.replay = replay
    _lib.TIMER = None
    tot, cnt = kt.summary()
    from gotennet_amd import engine as _eng
    zero_first = _eng.zero_X_in(rep.config(), 0)
    fam = {}
    for tag, t in tot.items():
        fam[family(tag)] = fam.get(family(tag), 0.0) + t
    dominant = max(fam, key=fam.get)               # kernel family with the largest share of the step
    stage_tags = {t for t in tot if t in MSG_STAGE}
    if a.breakdown and rank == 0:
        ssum = sum(tot.values())
        print(f"# {workload} b={B} lmax={lmax} mode={_engine_mode()}: per-kernel HIP-event breakdown of one step "
              f"({ssum:.3f} ms of events)", file=sys.stderr)
        for tag in sorted(tot, key=tot.get, reverse=True):
            print(f"  {tag:42s} {tot[tag]:8.3f} ms/step {cnt[tag]:3d} calls {1e3 * tot[tag] / cnt[tag]:8.1f} us/call "
                  f"{100 * tot[tag] / ssum:5.1f}%", file=sys.stderr)

    # ---- timed region: exactly K steps.  HIP events bracket the message-stage / HTR / message-backward launches on the
    # LAST `ev_steps` timed steps and the projection launches on the last one only: bracketing those ~23 launches on
    # every step cost 0.115 ms/step (8.11 vs 7.995 ms, same process), i.e. the measurement perturbed `value` by 1.4 %;
    # the per-launch averages are the same either way (111.1 us for the message stage in both).
    dom_tags = {t for t in tot if family(t) == dominant}
    always = stage_tags | {HTR_TAG, MSGB_TAG, "gn_htr_backward"}
    # (with batches in flight the bracketed steps run alone, i.e. slower than the rest: ONE such step -- 6 launches per stage,
    #  68 projection launches -- instead of three)
    ev_steps = min(1 if lanes > 1 else 3, steps)
    kt = KernelTimer(wanted=set())
    _lib.TIMER = kt
    lane_last = None
    fence()
    t0 = time.perf_counter()
    for it in range(steps):
        if it >= steps - ev_steps:                         # the event-bracketed steps run eagerly (a replay has no
            step_fn.replay = False                         # per-launch hooks): K - ev_steps replays + ev_steps eager steps
            kt.wanted = (dom_tags | always) if it == steps - 1 else set(always)
            if fl is not None and it == steps - ev_steps:
                fl.wait()                                  # the bracketed steps run one at a time: un-overlapped launch durations
                torch.cuda.current_stream().synchronize()
            e, f = step()
        else:
            e, f = step_lane() if fl is not None else step()
            if reducer is not None:
                lane_last = (e, reducer.bufs[(reducer.submitted - 1) % len(reducer.bufs)])
    fence()
    dt = time.perf_counter() - t0
    step_fn.replay = replay
    _lib.TIMER = None
    replayed = bool(replay and step_fn._graph_state is not None)
    rank_ms = None
    if dist is not None:
        # every rank's own time (a list in the record: the first multi-GPU run shows skew, not just the maximum), then the MAX
        tl = torch.zeros(world, dtype=torch.float64, device=dev)
        tl[rank] = dt
        dist.all_reduce(tl, op=dist.ReduceOp.SUM)
        rank_ms = [round(1e3 * float(v) / steps, 3) for v in tl.tolist()]
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(e).all() and torch.isfinite(f).all()
    # what the collective produced: every rank's shard present exactly once in the all-reduced vector
    n_ranks_seen = dist.get_world_size() if dist is not None else 1
    e_vec = e_all if dist is not None else e[:, 0]
    lanes_ok = True
    if dist is not None:
        assert torch.equal(e_all[rank * B:(rank + 1) * B], e[:, 0]), "all-reduced energy vector lost this rank's shard"
        # the same for the LAST laned step (its collective left from the reducer's communication stream), and the count of
        # non-zero entries of its vector = every rank's shard arrived in THAT collective; the verdict is shared by all ranks
        ok = 1.0
        if lane_last is not None:
            e_l, buf = lane_last
            ok = float(torch.equal(buf[rank * B:(rank + 1) * B], e_l[:, 0]) and int((buf != 0).sum()) == B * world)
        flag = torch.tensor([ok], dtype=torch.float32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        lanes_ok = bool(flag.item() == 1.0)
    energy_checksum = float(e_vec.double().sum())

    tot, cnt = kt.summary()

    dom_steps = 1

    def roof_gemm_family():
        """All projection launches of the timed region: algorithmic flops / summed launch time."""
        flops = t_ms = 0.0
        n = 0
        big = None
        for tag in tot:
            if family(tag) != "gn_gemm":
                continue
            fl = sum(2.0 * m_ * n_ * k_ for m_, n_, k_ in
                     (tuple(int(v) for v in part.rstrip("g").split("x")) for part in tag[8:-1].split("+")))
            flops += fl * cnt[tag]
            t_ms += tot[tag]
            n += cnt[tag]
            us = 1e3 * tot[tag] / cnt[tag]
            if big is None or tot[tag] > big[1]:
                big = (tag, tot[tag], us, fl / (us * 1e-6) / 1e12)
        ach = flops / (t_ms * 1e-3) / 1e12
        exact = _engine_mode() == "f32"
        common = dict(traffic=_pmc_traffic("gn_gemm_family_avg", lmax, workload, B),
                      us_per_launch=round(1e3 * t_ms / n, 2), launches_per_step=n // dom_steps,
                      algorithmic_flops_per_step=flops / dom_steps)
        if exact:
            return dict(kernel="gn::gemm_f32_mfma (all projection launches, exact fp32 MFMA)", bound="mfma",
                        achieved=round(ach, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(ach / MFMA_F32_PEAK_TF, 4),
                        **common,
                        largest_launch=dict(shape_MxNxK=big[0][8:-1], us=round(big[2], 2), tflops=round(big[3], 2),
                                            frac=round(big[3] / MFMA_F32_PEAK_TF, 4)))
        # split modes: the kernel EXECUTES several 16-bit MFMA flops per algorithmic fp32 flop (six bf16 terms, or three
        # fp16 terms with block exponents), so it is priced against the dense 16-bit matrix peak:
        # achieved = terms x algorithmic flops / time
        terms, kname, what = ((3, "gn::gemm_f16x2_mfma + gn::gemm_f16x2_colpipe + gn::gemm_f16x2_panel", "3 fp16 MFMAs on 2 scaled fp16 planes") if _engine_mode() == "f16x2"
                              else (6, "gn::gemm_bf16x3_mfma", "6 bf16 MFMAs on 3 bf16 planes"))
        return dict(kernel=f"{kname} (all projection launches; every fp32 product as {what}, fp32 accumulate)",
                    bound="mfma", achieved=round(terms * ach, 1), peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s",
                    frac=round(terms * ach / MFMA_BF16_PEAK_TF, 4),
                    note=f"achieved = EXECUTED 16-bit MFMA flops ({terms} x algorithmic) / summed launch time; peak = dense bf16 / "
                         "fp16 MFMA (2.5 PF).  The fp16 mode executes half the flops of the bf16 mode for the same product: "
                         "compare algorithmic_tflops across modes, not frac",
                    executed_terms_per_product=terms,
                    algorithmic_tflops=round(ach, 2), algorithmic_vs_fp32_mfma_peak=round(ach / MFMA_F32_PEAK_TF, 4),
                    **common,
                    largest_launch=dict(shape_MxNxK=big[0][8:-1], us=round(big[2], 2), executed_tflops=round(terms * big[3], 1),
                                        frac=round(terms * big[3] / MFMA_BF16_PEAK_TF, 4), algorithmic_tflops=round(big[3], 2)))

    def roof_message():
        """GATA message STAGE: SURVEY 8d B_msg over the summed duration of the stage's launches (one fused launch,
        or scores/softmax + message/aggregate)."""
        tags = sorted(t for t in tot if t in MSG_STAGE)
        layers = cnt[tags[0]] // ev_steps if tags else 0
        us = sum(1e3 * tot[t] / cnt[t] for t in tags)              # per layer: one launch of each stage kernel
        full = algorithmic_bytes_message(N, E, F, M, D)
        first = algorithmic_bytes_message(N, E, F, M, D, first_nd=lmax) if zero_first and layers else None
        nbytes = (full * (layers - 1) + first) / layers if first else full      # launch-weighted mean over the layers
        ach = nbytes / (us * 1e-6) / 1e9
        out = dict(kernel="+".join(tags), bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                   frac=round(ach / HBM_PEAK_GBS, 4), traffic=_pmc_traffic("message_stage", lmax, workload, B),
                   us_per_launch=round(us, 2), us_by_kernel={t: round(1e3 * tot[t] / cnt[t], 2) for t in tags},
                   launches_per_step=layers, algorithmic_bytes_per_launch=int(nbytes))
        if first:
            out.update(algorithmic_bytes_general_launch=full, algorithmic_bytes_first_launch=first,
                       note="the first interaction starts from X = 0: its launch skips the tensor-gate blocks and the X_in "
                            "table and is priced with its own, smaller byte count; bytes and time are means over the "
                            f"{layers} launches of a step")
            tot_s, cnt_s = kt.summary(split_first=True)
            k6, k6f, sm = "gn_message_aggregate", "gn_message_aggregate|first", "gn_attn_softmax"
            if k6 in tot_s and k6f in tot_s and sm in tot_s:
                us_sm = 1e3 * tot_s[sm] / cnt_s[sm]
                us_g, us_f = us_sm + 1e3 * tot_s[k6] / cnt_s[k6], us_sm + 1e3 * tot_s[k6f] / cnt_s[k6f]
                out.update(general_launches=dict(us_per_launch=round(us_g, 2),
                                                 frac=round(full / (us_g * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)),
                           first_launch=dict(us_per_launch=round(us_f, 2),
                                             frac=round(first / (us_f * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)))
        return out

    def roof_htr():
        """K7 gn_htr_edge: the kernel's own compulsory bytes (EQ / EK tables, rl, index, w written) / its duration."""
        if HTR_TAG not in tot:
            return None
        us = 1e3 * tot[HTR_TAG] / cnt[HTR_TAG]
        nbytes = 4 * N * 2 * D * F + E * (4 * (F + D) + 16)
        ach = nbytes / (us * 1e-6) / 1e9
        return dict(kernel=HTR_TAG, bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 4), traffic=_pmc_traffic(HTR_TAG, lmax, workload, B), us_per_launch=round(us, 2),
                    launches_per_step=cnt[HTR_TAG] // ev_steps, algorithmic_bytes_per_launch=nbytes,
                    note="bytes = 4N*2DF (EQ, EK tables) + E(4(F + D) + 16) (w written, rl, 2 x int64 index): the kernel's "
                         "own share of SURVEY 8d B_htr (the t read / t' write of the stage sit in the gated GEMM epilogue)")

    def roof_msg_backward():
        """gn_message_backward (by-target + by-source passes of one layer, timed together): compulsory bytes / duration."""
        if MSGB_TAG not in tot:
            return None
        us = 1e3 * tot[MSGB_TAG] / cnt[MSGB_TAG]
        layers = cnt[MSGB_TAG] // ev_steps
        nbytes = algorithmic_bytes_message_backward(N, E, F, M, D, H)
        if zero_first and layers:                                    # the first interaction's launch pair moves less
            nbytes = int((nbytes * (layers - 1) + algorithmic_bytes_message_backward_first(N, E, F, lmax, D, H)) / layers)
        ach = nbytes / (us * 1e-6) / 1e9
        return dict(kernel=MSGB_TAG + " (target + source passes of one layer)", bound="hbm", achieved=round(ach, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                    traffic=_pmc_traffic(MSGB_TAG, lmax, workload, B), us_per_launch=round(us, 2),
                    launches_per_step=cnt[MSGB_TAG] // ev_steps, algorithmic_bytes_per_launch=nbytes,
                    note="bytes = eproj read ONCE + g_eproj written once + a, rl, cut, g_s, g_rl, g_cut + node tables "
                         "(x, v, q|k, X_in, g_h1, g_X1 read; g_x, g_v, g_q|g_k, g_X written); the kernels read eproj in "
                         "both passes, so PMC traffic above this figure is the second read")

    def roof_htr_backward():
        """gn_htr_backward (by-target + by-source passes of one layer): compulsory bytes / duration."""
        tag = "gn_htr_backward"
        if tag not in tot:
            return None
        us = 1e3 * tot[tag] / cnt[tag]
        nbytes = 4 * E * (4 * F + 2 * D) + 16 * E + 4 * N * 4 * D * F
        ach = nbytes / (us * 1e-6) / 1e9
        return dict(kernel=tag + " (target + source passes of one layer)", bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=_pmc_traffic(tag, lmax, workload, B),
                    us_per_launch=round(us, 2), launches_per_step=cnt[tag] // ev_steps, algorithmic_bytes_per_launch=nbytes,
                    note="bytes = 4E(4F + 2D) + 16E + 4N*4DF: g_t', pre_t, w read and g_pre_t written [E,F]; rl read, g_rl written "
                         "[E,D]; 2 x int64 index; EQ, EK read and g_EQ, g_EK written [N,D,F]")

    def roof_gated_gemm():
        """The HTR edge update t' = t + SiLU(W_t t + b) * w as ONE projection launch (gotennet.py:611): HBM-priced, it moves
        four [E,F] streams (t read -- operand and residual are the same rows --, w read, t' and the pre-activation written)
        around a [E x F x F] product."""
        tags = [t for t in tot if family(t) == "gn_gemm" and any(part.endswith("g") for part in t[8:-1].split("+"))]
        if not tags:
            return None
        t_ms = sum(tot[t] for t in tags)
        n = sum(cnt[t] for t in tags)
        us = 1e3 * t_ms / n
        nbytes = 4 * E * F * 4 + 2 * 4 * F * F
        ach = nbytes / (us * 1e-6) / 1e9
        fl = 2.0 * E * F * F
        return dict(kernel="gn_gemm gated residual launch " + tags[0][7:], bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, us_per_launch=round(us, 2),
                    launches_per_step=n // dom_steps, algorithmic_bytes_per_launch=nbytes,
                    algorithmic_tflops=round(fl / (us * 1e-6) / 1e12, 1),
                    note="bytes = 4 x 4EF (t read once: operand AND residual, w read, t' written, pre-activation written for the "
                         "backward) + the weight planes; the rider product of the launch (gamma_m.0, atom-sized) is not priced")

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
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "split": "f32 (3xbf16-split MFMA, fp32 accumulate)",
                      "f16x2": "f32 (2xfp16-split MFMA with block exponents, fp32 accumulate)"}[_engine_mode()], "data": "synthetic",
            "config": {"workload": f"{workload} batch={B}/GPU (N={N} atoms, E={E} edges incl. self-loops), "
                                   f"n_atom_basis={F}, n_interactions={L}, lmax={lmax}, n_rbf={R}, heads={H}, "
                                   "sep_dir/sep_tensor, energy+forces",
                       "global_batch": B * world, "parallelism": f"dp{world} (molecule shards, 1 all-reduce)"},
            "n_ranks_seen": n_ranks_seen, "energy_vector_len": int(e_vec.numel()), "energy_checksum": energy_checksum,
            "rank_ms_per_step": rank_ms,
            "launch_mode": (f"hipGraph replay of the static-topology step ({steps - ev_steps} of {steps} timed steps; the last "
                            f"{ev_steps} run eagerly under HIP-event brackets)" if replayed else
                            (("eager launches; fresh batch every step (CSR / CSC / out-degree / molecule offsets rebuilt inside the timed step)"
                              if fresh else "eager launches; static topology (index arrays cached across steps)")
                             + (f"; {lanes} batches in flight on {lanes} HIP streams for the first {steps - ev_steps} timed steps, the last "
                                f"{ev_steps} (HIP-event brackets) one at a time" if lanes > 1 else "; one step at a time"))),
            "batches_in_flight": lanes, "lanes_consistent": lanes_ok,
            "roofline": roof_gemm_family() if dominant == "gn_gemm" else
            (roof_message() if dominant in MSG_STAGE else roof_other(dominant)),
            "roofline_gather_scatter": roof_message(),
            "roofline_htr_edge": roof_htr(),
            "roofline_message_backward": roof_msg_backward(),
            "roofline_htr_backward": roof_htr_backward(),
            "roofline_gated_gemm": roof_gated_gemm(),
        }
        if LIVE_TRAFFIC.get("_key") == (workload, B, lmax) and "mfma_busy" in LIVE_TRAFFIC and out["roofline"].get("bound") == "mfma":
            out["roofline"]["mfma_busy"] = LIVE_TRAFFIC["mfma_busy"]
        # the stage fractions again INSIDE `roofline` (a consumer that keeps only that object still sees them)
        out["roofline"]["stages"] = {k: ({kk: out[k][kk] for kk in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                                       "us_per_launch", "algorithmic_bytes_per_launch", "traffic")}
                                         if out[k] else None)
                                     for k in ("roofline_gather_scatter", "roofline_htr_edge", "roofline_message_backward")}
        live = LIVE_TRAFFIC.get("_key") == (workload, B, lmax)
        out["roofline"]["traffic_source"] = (
            "measured in this run: two rocprofv3 --kernel-trace --pmc sub-runs of this script (FETCH_SIZE, WRITE_SIZE; "
            "2 x FETCH + WRITE KiB per the gfx950 note), per-launch averages" if live else
            "committed rocprofv3 --pmc passes of this workload (profiles/pmc_traffic.json, FETCH_SIZE x2 + WRITE_SIZE per "
            "the gfx950 note), not re-measured in this run")
    return {"out": out, "rep": rep, "head": head, "lanes": lanes, "lanes_ok": lanes_ok, "n_ranks_seen": n_ranks_seen}


#: HBM bytes per launch measured IN THIS RUN (tag -> bytes), filled by live_traffic() before the headline record is built
LIVE_TRAFFIC = {}


def _pmc_counter_table(db_path, counter):
    """kernel name -> (average counter value per dispatch, dispatches) from a rocprofv3 rocpd database."""
    import sqlite3
    from collections import defaultdict
    cur = sqlite3.connect(db_path).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    acc = defaultdict(lambda: [0.0, 0])
    for k, cn, v in cur.execute(f"select {kcol}, counter_name, value from counters_collection"):
        if cn == counter:
            acc[k][0] += v
            acc[k][1] += 1
    return {k: (sm / n, n) for k, (sm, n) in acc.items() if n}


def _traffic_entry(f, w):
    """Per-launch HBM bytes of the kernel families from the FETCH_SIZE / WRITE_SIZE tables of one workload:
    (2 x FETCH_SIZE + WRITE_SIZE) KiB -- the counters are in KiB and on gfx950 FETCH_SIZE reports half the bytes of wide
    coalesced reads (MI355X_MICROARCH.md, HBM section; tools/pmc_traffic.py applies the same rule to the committed passes)."""
    byt = lambda k: int((2 * f[k][0] + w.get(k, (0.0, 0))[0]) * 1024)
    msg = [k for k in f if "message_aggregate" in k]
    soft = [k for k in f if "attn_softmax" in k]
    htr = [k for k in f if "htr_edge" in k]
    gem = [k for k in f if "gn::gemm_" in k]
    mb = [k for k in f if "msg_bwd_" in k or "attn_bwd_kernel" in k]
    if not (msg and soft and gem):
        return {}
    layers = f[soft[0]][1]                       # the softmax runs once per interaction: launches = layers x steps
    out = {"gn_message_aggregate": int(sum(byt(k) * f[k][1] for k in msg) / layers), "gn_attn_softmax": byt(soft[0]),
           "gn_gemm_family_avg": int(sum(byt(k) * f[k][1] for k in gem) / sum(f[k][1] for k in gem))}
    if htr:
        out["gn_htr_edge"] = int(sum(byt(k) * f[k][1] for k in htr) / max(f[k][1] for k in htr))
    if mb:
        out["gn_message_backward"] = int(sum(byt(k) * f[k][1] for k in mb) / layers)
    hb = [k for k in f if "htr_bwd_" in k]
    if hb:                                       # target + source passes of one layer (the last layer has no HTR stage)
        out["gn_htr_backward"] = int(sum(byt(k) * f[k][1] for k in hb) / max(f[k][1] for k in hb if "target" in k))
    out["message_stage"] = out["gn_message_aggregate"] + out["gn_attn_softmax"]
    return out


def live_traffic(a):
    """`roofline.traffic` measured in this run: two rocprofv3 sub-runs of THIS script on the headline workload (separate
    --pmc passes for FETCH_SIZE and WRITE_SIZE, --kernel-trace only: the combination the guide prescribes), two steps each,
    read back from the rocpd databases.  Returns {} -- and the record falls back to the committed passes, labelled so --
    when rocprofv3 is missing, fails or times out."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {}
    tabs = {}
    try:
        with tempfile.TemporaryDirectory(prefix="gn_pmc_", dir="/tmp") as td:
            for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"):
                out = os.path.join(td, counter.split()[0])
                cmd = [exe, "--kernel-trace", "--pmc", *counter.split(), "-d", out, "-o", "r", "--", sys.executable,
                       os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", str(a.batch),
                       "--lmax", str(a.lmax), "--workload", a.workload, "--no-lmax4", "--no-split", "--no-graph",
                       "--no-workloads", "--no-cpu-baseline", "--no-forward-only", "--no-live-traffic", "--no-static", "--lanes", "1"]
                env = dict(os.environ, TMPDIR="/tmp")
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=240, check=True)
                dbs = [os.path.join(r, fn) for r, _, fns in os.walk(out) for fn in fns if fn.endswith("_results.db")]
                if not dbs:
                    return {}
                for cn in counter.split():
                    tabs[cn] = _pmc_counter_table(dbs[0], cn)
        ent = _traffic_entry(tabs["FETCH_SIZE"], tabs["WRITE_SIZE"])
        # matrix-pipe occupancy of the projection family INSIDE the step: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of
        # the 1024 SIMDs, GRBM_GUI_ACTIVE the active cycles of the 8 XCDs (128 SIMDs each)
        mb, ga = tabs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}), tabs.get("GRBM_GUI_ACTIVE", {})
        gem = [k for k in mb if "gn::gemm_" in k and k in ga]
        if ent and gem:
            busy = sum(mb[k][0] * mb[k][1] for k in gem)
            act = sum(ga[k][0] * ga[k][1] for k in gem)
            ent["mfma_busy"] = round(busy / (128.0 * act), 4) if act else None
        return ent
    except Exception as exc:                                   # noqa: BLE001 -- a measurement aid must never fail the bench
        print(f"# live traffic pass failed ({type(exc).__name__}: {exc}); using the committed PMC passes", file=sys.stderr)
        return {}


def _pmc_traffic(tag, lmax, workload="rmd17_aspirin", batch=128):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), if present for this
    workload, batch and lmax."""
    if LIVE_TRAFFIC.get("_key") == (workload, batch, lmax) and tag in LIVE_TRAFFIC:
        return LIVE_TRAFFIC[tag]
    key = f"lmax{lmax}" if (workload == "rmd17_aspirin" and batch == 128) else f"{workload}_b{batch}_lmax{lmax}"
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return d.get(key, {}).get(tag)
    except Exception:
        return None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _physical_cores():
    """Physical cores of the host (unique (socket, core) pairs of /proc/cpuinfo; logical count / 2 when that is unreadable)."""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":", 1)[1].strip()))
        if pairs:
            return len(pairs)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1048576.0
    except Exception:
        pass
    return 32.0


def cpu_baseline(rep, head, workload, lmax, forces=True):
    """The CPU oracle (checker) timed on this box's host cores (BASELINE.md section 3 protocol).  A thread sweep
    (1 / 16 / 64 / all physical cores) on an 8-molecule sample picks the thread count; the figure is then measured at that
    count on the LARGEST batch the bound allows: energy+forces (torch autograd) on up to 64 molecules -- what fits both the
    host RAM (~0.6 GB of autograd state per molecule at lmax 2, the reference itself needs > 62 GB for 128) and ~10 s of CPU
    work --, energy only on the full batch of the headline.  1 warm-up, then one timed run per configuration; the sweep
    table is part of the record."""
    from gotennet_amd import synthetic
    from oracle import gotennet_oracle as orc
    phys = _physical_cores()
    sd = {k: v.detach().cpu() for k, v in rep.state_dict().items()}
    hsd = {k: v.detach().cpu() for k, v in head.state_dict().items()}
    c = rep.config()
    cfg = orc.default_config(n_atom_basis=c.F, n_interactions=c.L, n_rbf=c.R, num_heads=c.H, scale_edge=c.scale_edge,
                             lmax=lmax, sep_dir=c.sep_dir, sep_tensor=c.sep_tensor, cutoff=c.cutoff)

    def run_once(z, pos, batch, nm):
        if forces:
            return orc.energy_and_forces(sd, cfg, hsd, z, pos, batch, nm)
        with torch.no_grad():                                            # energy only: forward + head, no autograd graph
            ei, w, vec = orc.distance(pos, batch, cfg["cutoff"])
            h, _ = orc.gotennet_forward(sd, cfg, z, ei, w, vec)
            return orc.atomwise_energy(hsd, h, batch, nm, z=z)

    def timed(nm, nthreads, nruns=1, warm=True):
        torch.set_num_threads(nthreads)
        pos, batch, z = synthetic.make_batch(workload, nm, seed=0)
        if warm:
            run_once(z, pos.clone(), batch, nm)
        ts = []
        for _ in range(nruns):
            t0 = time.perf_counter()
            run_once(z, pos.clone(), batch, nm)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]

    old_threads = torch.get_num_threads()
    try:
        sweep = {}
        for nt in sorted({1, min(16, phys), min(64, phys), phys}):
            sweep[nt] = round(8 / timed(8, nt, nruns=2), 2)
        best_nt = max(sweep, key=sweep.get)
        n_atoms = synthetic.WORKLOADS[workload][0]
        per_mol_gb = 0.6 * (n_atoms / 21.0) * (((lmax + 1) ** 2 - 1) / 8.0 + 1.0) / 2.0
        nm_big = 128 if not forces else int(max(8, min(64, _mem_available_gb() * 0.4 / per_mol_gb)))
        nm_big = min(nm_big, synthetic.WORKLOADS[workload][2] if workload != "rmd17_aspirin" else 128)
        t_big = timed(nm_big, best_nt, nruns=1, warm=False)              # (the sweep warmed the allocator and the thread pool)
    finally:
        torch.set_num_threads(old_threads)
    val_big, val_sweep = nm_big / t_big, sweep[best_nt]
    value, nm_used = (val_big, nm_big) if val_big >= val_sweep else (val_sweep, 8)     # report the best, as BASELINE.md asks
    what = "energy+forces (torch autograd)" if forces else "energy only (forward + head, no_grad)"
    return {"value": round(value, 2), "unit": "molecules/s", "cores": best_nt, "kind": "port",
            "cpu_model": _cpu_model(), "host_cores_logical": os.cpu_count(), "host_cores_physical": phys, "torch": torch.__version__,
            "thread_sweep_8_molecules": {str(k): v for k, v in sweep.items()},
            "largest_batch": {"molecules": nm_big, "value": round(val_big, 2), "threads": best_nt},
            "sample_short": f"{nm_used} molecules of {workload}, {what}, CPU oracle port, best of a 1/16/64/{phys}-thread sweep: {best_nt} threads",
            "sample": f"{nm_used} molecules of {workload} (same model: F=256, L=6, lmax={lmax}), {what} on the CPU oracle "
                      f"(oracle/gotennet_oracle.py); thread sweep on 8 molecules {sweep} molecules/s, then {nm_big} molecules at "
                      f"{best_nt} threads: {val_big:.2f} molecules/s; the better of the two is `value`"}


if __name__ == "__main__":
    main()
