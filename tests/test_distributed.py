"""CPU, world_size 2, gloo: the molecule sharding + single energy all-reduce used for N > 1 GPUs.
(The per-rank arithmetic here is the CPU oracle: this test covers the distributed logic only.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _energies(first, count):
    from gotennet_amd import synthetic
    from oracle import gotennet_oracle as orc
    from tests.golden_util import load_case
    cfg, sd, head, _ = load_case("l2_sep_f32")
    pos, batch, z = synthetic.make_batch("qm9_small", count, seed=0, first_molecule=first)
    z = z.clamp(max=cfg["max_z"] - 1)
    e, f, _ = orc.energy_and_forces(sd, cfg, head, z, pos, batch, count)
    return e.reshape(-1), f


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gotennet_amd.parallel import reduce_energies, shard_range
    torch.set_num_threads(1)
    first, count = shard_range(rank, world, total)
    e_local, _ = _energies(first, count)
    e_all = reduce_energies(e_local, first, total)
    q.put((rank, first, count, e_all))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_batch():
    from gotennet_amd.parallel import shard_range
    for world in (1, 2, 3, 8):
        for total in (1, 5, 128, 1024, 1027):
            got = [shard_range(r, world, total) for r in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == total
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f0 + c0 == f1


def test_two_rank_energy_allreduce_matches_single_process():
    total, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    e_ref, _ = _energies(0, total)                # one process, whole batch
    for rank, first, count, e_all in res:
        assert e_all.shape == (total,)
        assert torch.allclose(e_all, e_ref, rtol=1e-5, atol=1e-6)   # molecules are independent units


def _bench_selftest(gpus, batch):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--selftest-dist",
                          "--batch", str(batch), "--steps", "3", "--workload", "qm9_small"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launches_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no launcher: bench.py spawns the ranks itself (torch.multiprocessing.spawn), they
    rendezvous on 127.0.0.1, shard the molecules, all-reduce the zero-padded per-molecule vector and rank 0 prints one
    JSON line.  (--selftest-dist: gloo, the step replaced by a checksum of the synthetic inputs; no kernel runs.)"""
    from gotennet_amd import synthetic
    B = 3
    two = _bench_selftest(2, B)
    assert two["n_ranks_seen"] == 2 and two["global_batch"] == 2 * B and two["energy_vector_len"] == 2 * B
    assert two["shards_consistent"] is True
    one = _bench_selftest(1, B)
    assert one["n_ranks_seen"] == 1 and one["energy_vector_len"] == B
    # rank r owns molecules [B r, B (r + 1)): the 2-rank vector is the first 2B molecules of the workload
    pos, batch, z = synthetic.make_batch("qm9_small", 2 * B, seed=0)
    val = torch.zeros(2 * B, dtype=torch.float64).index_add_(0, batch, z.double() * (pos.double() ** 2).sum(1)).float()
    assert abs(two["energy_checksum"] - float(val.double().sum())) < 1e-3 * abs(float(val.double().sum()))
    assert abs(one["energy_checksum"] - float(val[:B].double().sum())) < 1e-3 * abs(float(val[:B].double().sum()))


def test_bench_eight_ranks_as_the_driver_launches_them():
    """The exact command line of the 8-GPU scaling run -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W` -- with --selftest-dist (gloo, no
    kernel): eight ranks rendezvous, every rank's shard lands exactly once in the all-reduced vector, ONE JSON line."""
    import json
    import subprocess
    import sys
    from gotennet_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    B, world = 2, 8
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--selftest-dist", "--batch", str(B),
           "--workload", "qm9_small"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_ranks_seen"] == world and d["global_batch"] == world * B and d["energy_vector_len"] == world * B
    assert d["shards_consistent"] is True
    pos, batch, z = synthetic.make_batch("qm9_small", world * B, seed=0)
    val = torch.zeros(world * B, dtype=torch.float64).index_add_(0, batch, z.double() * (pos.double() ** 2).sum(1)).float()
    assert abs(d["energy_checksum"] - float(val.double().sum())) < 1e-3 * abs(float(val.double().sum()))
    # and the self-launching form the driver falls back to: `python bench.py --gpus 8`
    eight = _bench_selftest(world, B)
    assert eight["n_ranks_seen"] == world and abs(eight["energy_checksum"] - d["energy_checksum"]) < 1e-6 * abs(d["energy_checksum"])


def _ordered_worker(rank, world, port, B, steps, lanes, q):
    """2 ranks x 3 lanes: the bench's laned step with CPU tensors standing in for the energies.  Each rank drives its lanes
    round-robin; a lane's "energies" of step s are s-dependent, so a collective that paired step s of one rank with step
    s' != s of the other would show in the sums."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gotennet_amd.parallel import OrderedReducer
    red = OrderedReducer(B * world, rank * B, "cpu", slots=lanes)
    seen = []
    for s in range(steps):
        if (s + rank) % 3 == 0:
            time.sleep(0.01)                          # the ranks drift against each other
        e_local = torch.arange(B, dtype=torch.float32) + 100.0 * s + 1000.0 * rank + 1.0
        out = red.submit(e_local)
        assert out is red.bufs[s % lanes]
        seen.append(out.clone())
    red.wait()
    q.put((rank, [tuple(o) for o in red.order], torch.stack(seen)))
    dist.barrier()
    dist.destroy_process_group()


def test_ordered_reducer_two_ranks_three_lanes():
    """for world > 1 every lane's all-reduce leaves in SUBMISSION order from one place
    (parallel.OrderedReducer; on a GPU: one communication stream that waits on the lane's stream).  gloo, CPU tensors."""
    B, world, steps, lanes = 4, 2, 7, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ordered_worker, args=(r, world, port, B, steps, lanes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.stack([torch.cat([torch.arange(B, dtype=torch.float32) + 100.0 * s + 1000.0 * r + 1.0 for r in range(world)])
                        for s in range(steps)])
    for rank, order, seen in res:
        assert order == [(s, s % lanes) for s in range(steps)]
        assert torch.equal(seen, want)                # step s of rank 0 met step s of rank 1, every shard exactly once
