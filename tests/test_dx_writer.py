"""DX dump against text produced by the reference's own writer (molvoxel/etc/pymol/dx.py)."""
import numpy as np

from molvoxel_amd.etc.dx import format_dx, write_grid_to_dx_file
from tests import goldens


def test_dx_text_matches_reference(tmp_path):
    z = np.load(goldens.GOLD + "/dx_cases.npz")
    for name in ("a", "b", "c"):
        v, c, r = z[f"{name}/values"], z[f"{name}/center"], float(z[f"{name}/resolution"])
        ref = str(z[f"{name}/text"])
        assert format_dx(v, c, r) == ref
        p = tmp_path / f"{name}.dx"
        write_grid_to_dx_file(str(p), v, tuple(c), r)
        assert p.read_text() == ref
