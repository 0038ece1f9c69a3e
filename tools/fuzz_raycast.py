#!/usr/bin/env python3
"""Random raycast scenes against the CPU oracle, every tracking count: grid shape (sides 1 - 200, so the table's 16-voxel
window wraps or does not), voxel size, sensor pose (random rotation; inside, beside and far from the grid), range law,
max range, cloud size on either side of the direction sort's threshold, NaN / infinite points, float and double kernels,
workgroup sizes.  Prints one line per case that differs and a summary; exit code 1 on any difference.

    python tools/fuzz_raycast.py [--cases 300] [--seed 1]        (on the GPU box)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rotation(rng, translation):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    m = np.eye(4)
    m[:3, :3] = q
    m[:3, 3] = translation
    return m.T.reshape(16).copy()


def explain(ctx, O, kind, call, counts):
    """Bisects a differing cloud down to one ray and prints it with everything needed to replay it."""
    pts = call[0]
    run_gpu = (lambda p: _gpu(ctx, kind, (p,) + tuple(call[1:]), counts))
    run_cpu = (lambda p: (O.raycast_f64 if kind == "f64" else O.raycast_f32)(p, *call[1:]))
    idx = np.arange(len(pts))
    while len(idx) > 1:
        half = idx[:len(idx) // 2]
        if not np.array_equal(run_gpu(pts[half]), run_cpu(pts[half])):
            idx = half
        else:
            rest = idx[len(idx) // 2:]
            if np.array_equal(run_gpu(pts[rest]), run_cpu(pts[rest])):
                print("  (the difference needs rays of both halves: not a single ray's)")
                return
            idx = rest
    one = pts[idx]
    got, want = run_gpu(one), run_cpu(one)
    diff = np.argwhere(got != want)
    print("  ray %d: point %s = %s" % (idx[0], one[0].tolist(), [float(v).hex() for v in one[0]]))
    print("  max_range %r xform %s" % (call[1], [float(v).hex() for v in call[2]]))
    print("  voxel %s inverse %s sizes %s counts %s" % (float(call[3]).hex(), float(call[4]).hex(),
                                                       [float(v).hex() for v in call[5]], counts))
    for d in diff[:6]:
        print("  cell %s: gpu %s oracle %s" % (d.tolist(), got[tuple(d)].tolist() if got.ndim > 3 else got[tuple(d)], want[tuple(d)]))
    print("  visits gpu %d oracle %d" % (int(got.sum()), int(want.sum())), flush=True)


def _gpu(ctx, kind, call, counts):
    grids = ctx.tracking_grids(int(np.prod(counts)), 1)
    (grids.raycast_f64 if kind == "f64" else grids.raycast_f32)(0, *call)
    out = grids.retrieve(0, counts)
    grids.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--explain", action="store_true", help="bisect every differing cloud down to one ray and print it")
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", default="-1,64,256,1024", help="HIP_THREADS_PER_BLOCK values to draw from (-1 = not given)")
    args = ap.parse_args()
    from voxelized_geometry_tools_amd import capi, synthetic
    from oracle import oracle as O
    rng = np.random.default_rng(args.seed)
    contexts = {t: capi.Context(0, t) for t in (int(v) for v in args.threads.split(","))}
    bad = 0
    visits = 0
    t0 = time.time()
    for case in range(args.cases):
        sides = [int(rng.choice([1, 2, 7, 16, 17, 33, 64, 100, 150, 200])) for _ in range(3)]
        while np.prod(sides) > 4_000_000:
            sides[int(rng.integers(0, 3))] //= 2
        counts = tuple(max(1, s) for s in sides)
        vs = np.float32(rng.uniform(0.01, 0.2))
        npts = int(rng.choice([100, 12_000, 20_000, 40_000, 150_000]))
        pts = synthetic.raycast_cloud(npts, seed=int(rng.integers(1, 1 << 30)), nan_every=int(rng.choice([0, 7, 100])))
        pts = (pts * np.float32(rng.choice([0.02, 0.5, 1.0, 3.0, 20.0]))).astype(np.float32)
        if rng.random() < 0.2:
            pts[rng.integers(0, npts, 5)] = np.inf
        if rng.random() < 0.3:   # axis-aligned and diagonal rays: ties between the axes
            k = npts // 3
            pts[:k] = rng.choice([-1.0, 0.0, 1.0], (k, 3)).astype(np.float32) * np.float32(rng.uniform(0.1, 5.0))
        sizes = [np.float32(c) * vs for c in counts]
        where = [float(rng.choice([-3.0, -0.2, 0.0, 0.3, 0.5, 0.999, 1.0, 1.4])) * float(s) for s in sizes]
        xf = rotation(rng, where) if rng.random() < 0.7 else synthetic.translation_xform(*where)
        max_range = float(rng.choice([0.05, 0.7, 3.0, 10.0, 100.0]))
        threads = int(rng.choice(list(contexts)))
        ctx = contexts[threads]
        grids = ctx.tracking_grids(int(np.prod(counts)), 1)
        if rng.random() < 0.25:
            sizes64 = [float(c) * float(vs) for c in counts]
            args64 = (pts.astype(np.float64), max_range, xf.astype(np.float64), float(vs), 1.0 / float(vs), sizes64, counts)
            grids.raycast_f64(0, *args64)
            want = O.raycast_f64(*args64)
            kind = "f64"
        else:
            ivs = np.float32(1.0) / vs
            args32 = (pts, max_range, xf.astype(np.float32), vs, ivs, sizes, counts)
            grids.raycast_f32(0, *args32)
            want = O.raycast_f32(*args32)
            kind = "f32"
        got = grids.retrieve(0, counts)
        if args.explain and not np.array_equal(got, want):
            explain(ctx, O, kind, args64 if kind == "f64" else args32, counts)
        grids.close()
        visits += int(want.sum())
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH case %d: %s counts %s points %d threads %d max_range %g: %d cells differ" % (
                case, kind, counts, npts, threads, max_range, int(np.count_nonzero(got != want))), flush=True)
    for c in contexts.values():
        c.close()
    print("fuzz_raycast: %d cases, %d mismatches, %.1f M visits checked, %.0f s (seed %d)" % (
        args.cases, bad, visits / 1e6, time.time() - t0, args.seed))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
