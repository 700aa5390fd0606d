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
