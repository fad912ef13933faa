"""rocprofv3 helper: run one bench_configs row in a loop (python3 tools/run_cfg.py cfg4 128 50)."""
import sys

import numpy as np

sys.path.insert(0, ".")
import os

if os.environ.get("LIB"):  # another build of the library (A/B): path relative to the repo root
    from molvoxel_amd.voxelizer.hip import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["LIB"])
import bench_configs as bc
from molvoxel_amd import workloads as W

name, batch, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
wl = getattr(W, name)(batch=batch) if name != "cfg3" else W.cfg3()
print(bc.run(name, wl, list(range(batch)), steps))
