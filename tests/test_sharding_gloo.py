"""N > 1 path on CPU: two gloo ranks shard a batch of molecules exactly like bench.py / a multi-GPU job does.

The data path has no collective (molecules are independent); gloo is used only the way the real job uses RCCL:
a barrier and a MAX all-reduce of the elapsed time, plus here an all-gather of what each rank owned so the test can
check that the shards tile the batch without overlap. The per-rank "voxelization" on CPU is the oracle (checker
only): every rank's grids must equal the single-process result for the same molecules.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from molvoxel_amd import sharding
from molvoxel_amd import workloads as W


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle

    c_oracle.set_num_threads(1)
    wl = W.cfg4(batch=B, channels=4)
    lo, hi = sharding.shard_range(B, rank, world)
    offsets = np.cumsum([0] + [c.shape[0] for c in wl.coords])
    loc = sharding.local_offsets(offsets, lo, hi)
    assert loc[-1] == sum(c.shape[0] for c in wl.coords[lo:hi])
    sums = []
    for i in range(lo, hi):
        g = c_oracle.voxelize(wl.coords[i], wl.channels[i], 1.0, dimension=24, resolution=1.0)
        sums.append(float(g.sum(dtype=np.float64)))
    dist.barrier()
    t = torch.tensor([0.25 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    owned = [None] * world
    dist.all_gather_object(owned, (lo, hi, sums))
    if rank == 0:
        q.put((float(t.item()), owned))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    B, world = 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    tmax, owned = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 0.5  # MAX over ranks, as bench.py reports it
    covered = []
    for lo, hi, sums in owned:
        covered += list(range(lo, hi))
        assert len(sums) == hi - lo
    assert covered == list(range(B)), "shards must tile the batch exactly once"
    # single-process reference for the same molecules
    from oracle import c_oracle

    wl = W.cfg4(batch=B, channels=4)
    ref = [float(c_oracle.voxelize(wl.coords[i], wl.channels[i], 1.0, dimension=24, resolution=1.0).sum(dtype=np.float64))
           for i in range(B)]
    got = [s for _, _, sums in owned for s in sums]
    assert np.allclose(got, ref, rtol=0, atol=0)
