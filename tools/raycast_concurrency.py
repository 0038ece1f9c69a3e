#!/usr/bin/env python3
"""Do raycast calls on different streams overlap?  Config 3's cloud A, points resident: N calls into N tracking grids, all on
one stream against spread over S streams (one context per stream).  Prints ms per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from voxelized_geometry_tools_amd import capi, synthetic
    n, points, calls = 256, 1_000_000, 8
    counts = (n, n, n)
    vs = np.float32(5.12 / n)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    pts = torch.from_numpy(synthetic.raycast_cloud(points, seed=42)).cuda()
    xf = synthetic.translation_xform(2.56, 2.56, 2.56).astype(np.float32)
    for streams in (1, 2, 4):
        ctxs = [capi.Context(0) for _ in range(streams)]
        torch_streams = [torch.cuda.Stream() for _ in range(streams)]
        for c, s in zip(ctxs, torch_streams):
            c.set_stream(s.cuda_stream)
        grids = [c.tracking_grids(n ** 3, calls // streams) for c in ctxs]

        def run():
            for k in range(calls):
                grids[k % streams].raycast_f32_dev(k // streams, pts.data_ptr(), points, 3.0, xf, vs, ivs, sizes, counts)
            torch.cuda.synchronize()

        run()
        run()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            run()
            best = min(best, time.perf_counter() - t0)
        print("%d stream(s): %d calls in %.3f ms = %.3f ms per call" % (streams, calls, best * 1e3, best * 1e3 / calls), flush=True)
        for g in grids:
            g.close()
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
