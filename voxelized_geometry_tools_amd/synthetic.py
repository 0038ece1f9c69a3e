"""Seeded synthetic inputs for the hot path.

Pure numpy so the same grids can be produced in tests, in bench.py and in the
golden-fixture generator.  All randomness comes from a SplitMix64 stream so the
inputs do not depend on the numpy version.
"""
import numpy as np

_MASK = (1 << 64) - 1


class SplitMix64:
    """Sequential SplitMix64; scalar draws (used for small parameter lists)."""

    def __init__(self, seed=42):
        self.state = seed & _MASK

    def next_u64(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & _MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def uniform(self):
        return (self.next_u64() >> 11) * (1.0 / (1 << 53))


def splitmix64_array(seed, count):
    """Vectorised counter-mode SplitMix64: element i = mix(seed + (i+1)*gamma)."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_array(seed, count):
    return (splitmix64_array(seed, count) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def sphere_list(shape, seed=42, num_spheres=64):
    """D1 parameters: integer centres uniform in the grid, radii uniform in [2, max(n)/16].

    Returns int64 centres (num_spheres, 3) and float64 squared radii (num_spheres,).
    """
    rng = SplitMix64(seed)
    nx, ny, nz = shape
    rmax = max(2.0, max(shape) / 16.0)
    centres = np.zeros((num_spheres, 3), dtype=np.int64)
    r2 = np.zeros(num_spheres, dtype=np.float64)
    for i in range(num_spheres):
        centres[i] = (int(rng.uniform() * nx), int(rng.uniform() * ny), int(rng.uniform() * nz))
        r = 2.0 + rng.uniform() * (rmax - 2.0)
        r2[i] = r * r
    return centres, r2


def occupancy_spheres(shape, seed=42, num_spheres=64):
    """D1 "spheres": 1.0 inside 64 solid spheres, 0.0 outside."""
    nx, ny, nz = shape
    occ = np.zeros(shape, dtype=np.float32)
    centres, r2 = sphere_list(shape, seed, num_spheres)
    for (cx, cy, cz), rr in zip(centres, r2):
        r = int(np.ceil(np.sqrt(rr)))
        x0, x1 = max(cx - r, 0), min(cx + r + 1, nx)
        y0, y1 = max(cy - r, 0), min(cy + r + 1, ny)
        z0, z1 = max(cz - r, 0), min(cz + r + 1, nz)
        if x0 >= x1 or y0 >= y1 or z0 >= z1:
            continue
        dx = (np.arange(x0, x1, dtype=np.int64) - cx) ** 2
        dy = (np.arange(y0, y1, dtype=np.int64) - cy) ** 2
        dz = (np.arange(z0, z1, dtype=np.int64) - cz) ** 2
        d2 = dx[:, None, None] + dy[None, :, None] + dz[None, None, :]
        sub = occ[x0:x1, y0:y1, z0:z1]
        sub[d2.astype(np.float64) <= rr] = 1.0
    return occ


def occupancy_salt(shape, seed=42, p=0.01):
    """D2 "salt": i.i.d. Bernoulli(p) filled voxels."""
    n = int(np.prod(shape))
    u = uniform_array(seed, n)
    return (u < p).astype(np.float32).reshape(shape)


def occupancy_unknown_mix(shape, seed=42):
    """D3: D1 plus 1 % of the cells set to exactly 0.5 (unknown)."""
    occ = occupancy_spheres(shape, seed)
    u = uniform_array(seed + 1, occ.size).reshape(shape)
    occ[u < 0.01] = 0.5
    return occ


def occupancy_degenerate(shape, kind):
    """D4: 'empty', 'full' or 'single' (one filled voxel at the origin)."""
    if kind == "empty":
        return np.zeros(shape, dtype=np.float32)
    if kind == "full":
        return np.ones(shape, dtype=np.float32)
    if kind == "single":
        occ = np.zeros(shape, dtype=np.float32)
        occ[0, 0, 0] = 1.0
        return occ
    raise ValueError(kind)


def make_occupancy(shape, dist="spheres", seed=42):
    if dist == "spheres":
        return occupancy_spheres(shape, seed)
    if dist == "salt":
        return occupancy_salt(shape, seed)
    if dist == "unknown_mix":
        return occupancy_unknown_mix(shape, seed)
    return occupancy_degenerate(shape, dist)


def raycast_cloud(num_points=1_000_000, seed=42, nan_every=100):
    """C3 cloud: unit directions uniform on the sphere times a range
    uniform in [0.5, 4.0] m; every `nan_every`-th point is NaN.  float32 xyz AoS."""
    u = uniform_array(seed, 3 * num_points).reshape(num_points, 3)
    zc = 2.0 * u[:, 0] - 1.0
    phi = 2.0 * np.pi * u[:, 1]
    s = np.sqrt(np.maximum(0.0, 1.0 - zc * zc))
    rng = 0.5 + 3.5 * u[:, 2]
    pts = np.stack([s * np.cos(phi) * rng, s * np.sin(phi) * rng, zc * rng], axis=1)
    pts = pts.astype(np.float32)
    if nan_every:
        pts[::nan_every] = np.nan
    return pts


def translation_xform(tx, ty, tz):
    """Column-major 4x4 rigid transform with identity rotation (16 floats)."""
    m = np.eye(4, dtype=np.float64)
    m[:3, 3] = (tx, ty, tz)
    return m.T.reshape(16).copy()  # column-major flattening


def kernel_sources_sha256(group="edt"):
    """sha256 over the HIP sources a profile belongs to (csrc/edt_* + the shared headers, or the voxelizer's), in name
    order: profiles/*_current.json record it when they are collected and bench.py reports their numbers only for a tree
    whose sources still hash to it (the GPU box has no .git to ask)."""
    import glob
    import hashlib
    import os
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    patterns = {"edt": ["edt_*.hip", "edt_*.hpp", "vgt_internal.hpp"],
                "voxelizer": ["voxelizer_kernels.hip"]}[group]
    files = sorted(f for pat in patterns for f in glob.glob(os.path.join(csrc, pat)))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()
