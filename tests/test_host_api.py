"""Host-side logic without a GPU: operator contract, factory, transforms, C ABI surface, sharding."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import goldens

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_factory_and_contract_surface():
    import molvoxel_amd
    from molvoxel_amd.voxelizer.base import BaseRandomTransform, BaseVoxelizer
    from molvoxel_amd.voxelizer.hip import RandomTransform, Voxelizer

    assert issubclass(Voxelizer, BaseVoxelizer) and Voxelizer.LIB == "HIP"
    assert Voxelizer.transform_class is RandomTransform and issubclass(RandomTransform, BaseRandomTransform)
    assert BaseVoxelizer.RADII_TYPE_LIST == ["scalar", "channel-wise", "atom-wise"]
    assert BaseVoxelizer.DENSITY_TYPE_LIST == ["gaussian", "binary"]
    for name in ("forward", "forward_types", "forward_features", "forward_single", "get_empty_grid", "asarray",
                 "grid_dimension", "radii_type", "density_type", "resolution", "dimension", "width",
                 "spatial_dimension", "is_radii_type_scalar", "is_radii_type_channel_wise", "is_radii_type_atom_wise",
                 "is_density_type_binary", "is_density_type_gaussian"):
        assert hasattr(Voxelizer, name), name
    assert Voxelizer.__call__ is Voxelizer.forward
    # only the hip backend lives here; the reference's backends are refused loudly, never substituted
    for lib in ("numpy", "numba", "torch"):
        with pytest.raises(AssertionError):
            molvoxel_amd.create_voxelizer(library=lib)
    t = molvoxel_amd.create_random_transform(0.5, True)
    assert isinstance(t, RandomTransform) and t.random_translation == 0.5 and t.random_rotation is True


def test_base_geometry_and_property_semantics():
    """BaseVoxelizer state (reference base/voxelizer.py:15-97) on a do-nothing subclass."""
    from molvoxel_amd.voxelizer.base import BaseVoxelizer

    class Dummy(BaseVoxelizer):
        def forward_types(self, *a, **k): return "types"
        def forward_features(self, *a, **k): return "features"
        def forward_single(self, *a, **k): return "single"
        def get_empty_grid(self, *a, **k): return None
        def asarray(self, a, obj): return a

    v = Dummy(0.4, 64, "atom-wise", "gaussian", sigma=0.7)
    assert v.width == 0.4 * 63 and v.upper_bound == v.width / 2.0 and v.lower_bound == -v.upper_bound
    assert v.grid_dimension(5) == (5, 64, 64, 64) and v.spatial_dimension == (64, 64, 64)
    assert v._sigma == 0.7 and v.is_radii_type_atom_wise and v.is_density_type_gaussian
    v.density_type = "binary"
    assert v.is_density_type_binary
    v.density_type = "gaussian"  # the setter cannot carry sigma: back to the default 0.5 (reference quirk Q12)
    assert v._sigma == 0.5
    with pytest.raises(AssertionError):
        v.radii_type = "per-atom"
    v.density = "binary"  # what the reference's own tests do: creates an unrelated attribute (Q5)
    assert v.density_type == "gaussian"
    # dispatch on `channels` (reference base/voxelizer.py:121-128)
    assert v.forward(None, None, None, 1.0) == "single"
    assert v.forward(None, None, np.zeros(3, np.int16), 1.0) == "types"
    assert v(None, None, np.zeros((3, 2), np.float32), 1.0) == "features"


def test_transforms_match_reference_goldens_on_host():
    """Seeded RNG order and arithmetic of do_random_transform / T (reference numpy/transform.py) on numpy arrays."""
    from molvoxel_amd.voxelizer.hip.transform import RandomTransform, do_random_transform

    z, idx = goldens.load("transform_cases.npz")
    xyz, center = z["coords"], z["center"]
    keep = xyz.copy()
    for case in idx:
        np.random.seed(case["seed"])
        if case["id"].startswith("t"):
            out = do_random_transform(xyz, center if case["use_center"] else None, case["random_translation"],
                                      case["random_rotation"])
            assert np.array_equal(np.random.rand(2), z[f"{case['id']}/next_rand"]), "RNG draws consumed differ"
        else:
            T = RandomTransform(case["random_translation"], case["random_rotation"]).get_transform()
            out = T(xyz, center)
            if T.translation is not None:
                assert np.array_equal(T.translation, z[f"{case['id']}/translation"]) and T.translation.dtype == np.float32
            if T.quaternion is not None:
                assert np.array_equal(np.array(T.quaternion), z[f"{case['id']}/quaternion"])
        assert np.array_equal(out, z[f"{case['id']}/out"]), case["id"]
    assert np.array_equal(xyz, keep), "inputs must not be mutated"


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "mvx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvx_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    """No compute calls (no GPU here): the shared library must load and export all of include/mvx.h."""
    from molvoxel_amd.voxelizer.hip import _lib

    names = _declared_functions()
    assert len(names) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvx.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    loaded = _lib.load()
    assert loaded.mvx_version() == 140
    # struct layouts the ABI promises
    assert C.sizeof(_lib.MvxConfig) == 40 and C.sizeof(_lib.MvxXform) == 80


def test_no_gpu_fails_loudly_not_silently():
    """Without a HIP device the product path raises; it never falls back to a CPU implementation."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import molvoxel_amd

    with pytest.raises(RuntimeError, match="no usable HIP device"):
        molvoxel_amd.create_voxelizer()


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/: the package must not import or load it."""
    pat_py = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|libmvx_oracle|numpy_port|c_oracle", re.M)
    pat_c = re.compile(r"#include\s*[<\"].*oracle|libmvx_oracle|ovx_", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "molvoxel_amd")):
        for f in files:
            src = open(os.path.join(dirpath, f), errors="ignore").read() if f.endswith((".py", ".hip", ".h", ".cpp")) else ""
            pat = pat_py if f.endswith(".py") else pat_c
            assert not pat.search(src), f"{os.path.join(dirpath, f)} reaches into the oracle"


def test_workloads_are_deterministic_and_sized_like_baseline():
    from molvoxel_amd import workloads as W

    a, b = W.cfg2(), W.cfg2()
    assert np.array_equal(a.coords[0], b.coords[0]) and a.coords[0].shape == (4000, 3) and a.channels[0].shape == (4000, 32)
    assert a.algorithmic_bytes(0) == 4 * 32 * 64**3 + 4000 * (24 + 4 * 32 + 4)
    c3 = W.cfg3()
    assert c3.dimension == 48 and c3.density == "binary" and c3.channels[0].max() == 3
    c5 = W.cfg5()
    assert c5.dimension == 128 and c5.radii[0].dtype == np.float32 and 1.0 <= c5.radii[0].min() and c5.radii[0].max() < 2.0
    c4 = W.cfg4(batch=16)
    assert c4.batch == 16 and all(40 <= x.shape[0] <= 60 for x in c4.coords)


def test_shard_bounds_partition():
    from molvoxel_amd.sharding import balanced_shard_bounds, local_offsets, shard_bounds, shard_range

    for n in (0, 1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and (np.diff(b) >= 0).all() and np.diff(b).max() - np.diff(b).min() <= 1
            assert [shard_range(n, r, w) for r in range(w)] == [(int(b[r]), int(b[r + 1])) for r in range(w)]
    rng = np.random.default_rng(0)
    wts = rng.integers(40, 4000, 100)
    b = balanced_shard_bounds(wts, 8)
    assert b[0] == 0 and b[-1] == 100 and (np.diff(b) >= 0).all()
    loads = [wts[b[r]:b[r + 1]].sum() for r in range(8)]
    assert max(loads) <= wts.sum() / 8 + wts.max()
    off = np.cumsum(np.concatenate([[0], wts]))
    lo = local_offsets(off, 10, 20)
    assert lo[0] == 0 and lo[-1] == wts[10:20].sum() and len(lo) == 11


def test_contract_module_geometry_and_switches():
    """molvoxel_amd/voxelizer/contract.py against the reference's documented behaviour (base/voxelizer.py:15-97)."""
    from molvoxel_amd.voxelizer.contract import GridGeometry, VoxelizerContract

    geo = GridGeometry(0.5, 64)
    assert geo.width == 31.5 and geo.upper_bound == 15.75 and geo.lower_bound == -15.75
    assert geo.spatial_dimension == (64, 64, 64) and geo.grid_dimension(5) == (5, 64, 64, 64)

    class Dummy(VoxelizerContract):
        changed = 0

        def _density_changed(self):
            self.changed += 1

    v = Dummy(0.4, 24, "atom-wise", "gaussian", sigma=0.8)
    assert (v.resolution, v.dimension, v.width) == (0.4, 24, 0.4 * 23) and v._sigma == 0.8
    assert v.is_radii_type_atom_wise and not v.is_radii_type_scalar and not v.is_radii_type_channel_wise
    v.radii_type = "channel-wise"
    assert v.is_radii_type_channel_wise and v.radii_type == "channel-wise"
    with pytest.raises(AssertionError):
        v.radii_type = "per-atom"
    v.density_type = "binary"
    assert v.is_density_type_binary and v.changed == 1
    v.density_type = "gaussian"  # the setter cannot carry a sigma: back to the default (base/voxelizer.py:65-70)
    assert v.is_density_type_gaussian and v._sigma == 0.5 and v.changed == 2
    with pytest.raises(AssertionError):
        Dummy(0.5, 8, "scalar", "box")
    # forward dispatches on the channel argument
    calls = []
    v.forward_single = lambda *a: calls.append(("single", len(a)))
    v.forward_types = lambda *a: calls.append(("types", len(a)))
    v.forward_features = lambda *a: calls.append(("features", len(a)))
    v.forward(np.zeros((2, 3)), None, None, 1.0)
    v(np.zeros((2, 3)), None, np.zeros(2, int), 1.0)
    v.forward(np.zeros((2, 3)), None, np.zeros((2, 4)), 1.0, 0.5, True, None)
    assert calls == [("single", 6), ("types", 7), ("features", 7)]


def test_header_is_valid_c_and_matches_the_python_structs(tmp_path):
    """include/mvx.h must compile as plain C (the boundary is a C ABI) and agree with the ctypes layouts."""
    import shutil
    import subprocess

    from molvoxel_amd.voxelizer.hip import _lib

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "mvx.h"\n'
        "int main(void) {\n"
        '  printf("%zu %zu %zu %zu %zu %d\\n", sizeof(mvx_config), sizeof(mvx_xform), offsetof(mvx_config, precision),\n'
        "         offsetof(mvx_xform, trans), offsetof(mvx_xform, flags), MVX_VERSION);\n"
        "  return 0;\n}\n")
    exe = tmp_path / "abi"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert [int(v) for v in out] == [C.sizeof(_lib.MvxConfig), C.sizeof(_lib.MvxXform), _lib.MvxConfig.precision.offset,
                                     _lib.MvxXform.trans.offset, _lib.MvxXform.flags.offset, 140]


def _build_c_demo(tmp_path):
    import shutil
    import subprocess

    from molvoxel_amd.voxelizer.hip import _lib

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = tmp_path / "c_abi_demo"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call([gcc, "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", libdir, "-lmvx_hip",
                           f"-Wl,-rpath,{libdir}", "-lm", "-o", str(exe)])
    return str(exe)


def test_c_program_links_against_the_abi_and_fails_loudly_without_a_gpu(tmp_path):
    import subprocess

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
def test_c_program_runs_through_the_abi(tmp_path):
    """examples/c_abi_demo.c: plain C caller, device buffers from mvx_alloc, result against a brute-force host loop."""
    import subprocess

    r = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
