"""Times the REFERENCE numpy backend (imported from /root/reference, build container only) on the loop of its own
test/test_time_numpy.py as restated in bench_configs.harness (same inputs, 16 x 25 x 5 by default).

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference_harness.py [--trials 1]

Test infrastructure: prints seconds per run for single / types / features on this host's cores.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--trials", type=int, default=5)
    ap.add_argument("--iterations", type=int, default=25)
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.ref, "molvoxel")):
        sys.exit("reference not present")
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    from molvoxel.voxelizer.numpy import Voxelizer  # the reference backend (test_time_numpy.py:113-119)

    import bench_configs

    print(f"cores: {os.cpu_count()}")
    bench_configs.harness(Voxelizer(0.5, 48), num_iteration=args.iterations, num_trial=args.trials)


if __name__ == "__main__":
    main()
