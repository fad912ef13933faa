"""OpenDX text dump of one 3-D channel of a voxel grid.

Counterpart of the reference's `write_grid_to_dx_file` (molvoxel/etc/pymol/dx.py:2-39): same header lines, values
with five decimals, three per line, row-major order. Accepts numpy arrays or torch tensors (CUDA tensors are copied
to the host; this is an on-disk debugging format after the hot path, not part of it).
"""
from __future__ import annotations

import numpy as np


def format_dx(values, center, resolution) -> str:
    if hasattr(values, "detach"):
        values = values.detach().cpu().numpy()
    grid = np.asarray(values)
    assert grid.ndim == 3
    assert len(center) == 3
    nx, ny, nz = grid.shape
    origin = [float(c) - resolution * (n - 1) / 2.0 for c, n in zip(center, (nx, ny, nz))]
    head = [
        f"object 1 class gridpositions counts {nx:d} {ny:d} {nz:d}",
        "origin " + " ".join(f"{o:.5f}" for o in origin),
        f"delta {resolution:.5f} 0 0",
        f"delta 0 {resolution:.5f} 0",
        f"delta 0 0 {resolution:.5f}",
        f"object 2 class gridconnections counts {nx:d} {ny:d} {nz:d}",
        f"object 3 class array type double rank 0 items [ {nx * ny * nz:d} ] data follows",
    ]
    flat = [f"{v:.5f}" for v in grid.reshape(-1).tolist()]
    body = []
    for i in range(0, len(flat), 3):
        chunk = flat[i : i + 3]
        # a full triple ends with a newline; a trailing partial one keeps the reference's trailing blank
        body.append(" ".join(chunk) + ("\n" if len(chunk) == 3 else " "))
    return "\n".join(head) + "\n" + "".join(body)


def write_grid_to_dx_file(dx_path, values, center, resolution) -> None:
    with open(dx_path, "w") as fh:
        fh.write(format_dx(values, center, resolution))
