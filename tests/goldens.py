"""Helpers to read tests/golden/*.npz (written by oracle/gen_golden.py from the reference)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name))
    index = json.loads(str(z["index"])) if "index" in z.files else None
    return z, index


def small_case_inputs(z, case):
    g = f"g{case['geom']}"
    coords = z[f"{g}/coords"]
    mode = case["mode"]
    chan = {"features": z[f"{g}/features"], "types": z[f"{g}/types"], "single": None}[mode]
    rt = case["radii_type"]
    radii = {"scalar": case["scalar_radius"], "atom-wise": z[f"{g}/r_atom"], "channel-wise": z[f"{g}/r_chan"]}[rt]
    return coords, chan, radii
