"""One rank of the sharded cfg-4 job (run by tests/test_hip_multiprocess.py under torch.distributed.run).

Each rank voxelizes its atom-count-balanced shard of the ligand batch through the HIP path (one forward_batch on its
GPU - on a 1-GPU box both ranks share the device), compares every grid with the CPU oracle, and rank 0 checks that
the shards tile the batch. No data-path collective: gloo carries a barrier, a MAX of the elapsed time and the
gathered bookkeeping only, exactly as bench.py uses it.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    total = int(sys.argv[1])
    out_path = sys.argv[2]
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    import molvoxel_amd
    from molvoxel_amd import sharding
    from molvoxel_amd import workloads as W
    from oracle import c_oracle

    wl = W.cfg4(batch=total)
    bounds = sharding.balanced_shard_bounds([c.shape[0] for c in wl.coords], world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    coords, feats = wl.coords[lo:hi], wl.channels[lo:hi]
    offsets = sharding.local_offsets(np.cumsum([0] + [c.shape[0] for c in wl.coords]), lo, hi)
    vox = molvoxel_amd.create_voxelizer(0.5, 64, "scalar", "gaussian", library="hip", device=dev)
    d_coords, d_feats = vox.asarray(np.concatenate(coords), "coords"), vox.asarray(np.concatenate(feats), "features")
    dist.barrier()
    t0 = time.perf_counter()
    grid = vox.forward_batch(d_coords, offsets, None, d_feats, 1.0)
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    worst, sums = 0.0, []
    for b in range(hi - lo):
        ref = c_oracle.voxelize(coords[b], feats[b], 1.0, dimension=64)
        got = grid[b].cpu().numpy()
        assert np.array_equal(got != 0, ref != 0), f"rank {rank} molecule {lo + b}: membership differs"
        worst = max(worst, float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max()))
        sums.append(float(got.sum(dtype=np.float64)))
    assert worst <= 5e-6, worst
    owned = [None] * world
    dist.all_gather_object(owned, dict(rank=rank, lo=lo, hi=hi, atoms=int(offsets[-1]), worst=worst, sums=sums, device=dev))
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(dict(world=world, max_elapsed=float(el.item()), ranks=owned), fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
