"""Debug aid: default pipeline against the int16-field pipeline (variant 3) on small random grids; prints where they differ."""
import sys
import numpy as np
sys.path.insert(0, ".")
from voxelized_geometry_tools_amd import capi

rng = np.random.default_rng(5)
with capi.Context(0, testing=True) as ctx:
    for shape in [(1, 1, 12), (1, 2, 12), (4, 8, 12), (2, 3, 64), (2, 3, 65), (3, 5, 130), (2, 70, 200), (1, 1, 1100), (5, 4, 5000)]:
        occ = (rng.random(shape) < 0.3).astype(np.float32)
        ctx.set_edt_variant(3)
        want, _, _ = ctx.sdf_from_occupancy(occ, 0.25)
        ctx.set_edt_variant(0)
        got, _, _ = ctx.sdf_from_occupancy(occ, 0.25)
        bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        print(shape, "mismatches", len(bad))
        for idx in bad[:6]:
            x, y, z = idx
            print("   at", tuple(idx), "got", got[x, y, z], "want", want[x, y, z], "line", occ[x, y, max(0, z - 5):z + 6].astype(int))
