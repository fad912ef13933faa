"""bench.py against another build of the library (A/B runs on one box):  python3 tools/bench_lib.py <lib.so> [bench args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molvoxel_amd.voxelizer.hip import _lib

_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
