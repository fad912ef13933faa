"""Ligand + pocket of PDB 10GS -> one (14, 48, 48, 48) image on the GPU, then an OpenDX file per channel.

    python examples/voxelize_10gs.py [out_dir]

The same steps as the reference's README / test_run_*.py, without RDKit: read the molecules, pick channel getters,
wrap a voxelizer, run. Needs an MI355X (the HIP backend has no CPU path).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import molvoxel_amd  # noqa: E402
from molvoxel_amd.etc import mol as M  # noqa: E402


def main(out_dir="dx_10gs"):
    data = os.path.join(ROOT, "tests", "golden", "10gs")
    ligand = M.read_sdf(os.path.join(data, "10gs_ligand.sdf"))[0]
    pocket = M.read_pdb(os.path.join(data, "10gs_pocket_nowater.pdb"))
    atoms = M.AtomTypeGetter(["C", "N", "O", "S"], unknown=True)
    maker = M.ComplexPointCloudMaker(atoms, M.BondTypeGetter.default(), atoms, None, channel_type="types")

    voxelizer = molvoxel_amd.create_voxelizer(resolution=0.5, dimension=48, density_type="gaussian", library="hip")
    wrapper = M.ComplexWrapper(maker, voxelizer)
    center = ligand.coords.mean(axis=0)

    grid = wrapper.get_empty_grid()                                  # (14, 48, 48, 48) float32 on the GPU
    image = wrapper.run(ligand, pocket, center, radii=1.0, out_grid=grid)
    assert image is grid
    print("channels:", maker.channels)
    print("image:", tuple(image.shape), image.dtype, image.device, "sum per channel:",
          np.round(image.sum(dim=(1, 2, 3)).cpu().numpy(), 1))

    # data augmentation the reference's way: random rotation + translation drawn from numpy's global RNG
    np.random.seed(0)
    batch = wrapper.run_batch([[ligand, pocket]] * 8, centers=[center] * 8, radii=1.0, random_translation=0.5,
                              random_rotation=True)                  # 8 augmented copies, one launch
    print("augmented batch:", tuple(batch.shape))

    paths = wrapper.dump_dx(out_dir, image, center)
    print(f"wrote {len(paths)} DX files to {out_dir}/ (e.g. {os.path.basename(paths[0])})")


if __name__ == "__main__":
    main(*sys.argv[1:2])
