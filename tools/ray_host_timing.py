import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
from voxelized_geometry_tools_amd import capi, synthetic
ctx = capi.Context(0)
n = 256; counts = (n, n, n); vs = np.float32(5.12 / n); ivs = np.float32(1.0) / vs
sizes = [np.float32(c) * vs for c in counts]
pts = synthetic.raycast_cloud(1_000_000, seed=42)
xf = synthetic.translation_xform(2.56, 2.56, 2.56).astype(np.float32)
for rep in range(4):
    t0 = time.perf_counter(); grids = ctx.tracking_grids(n ** 3, 1); ctx.synchronize(); t1 = time.perf_counter()
    grids.raycast_f32(0, pts, 3.0, xf, vs, ivs, sizes, counts); t2 = time.perf_counter()
    env = np.zeros(counts, dtype=np.float32); env[:, :, 0] = 1.0
    t3 = time.perf_counter(); fg = ctx.filter_grid(env); t4 = time.perf_counter()
    fg.filter(grids, 1.0, 1, 1); ctx.synchronize(); t5 = time.perf_counter()
    out = fg.retrieve(); t6 = time.perf_counter()
    grids.close(); fg.close(); t7 = time.perf_counter()
    print("grids %.2f raycast(host pts) %.2f filter_grid(create+upload) %.2f filter %.2f retrieve %.2f close %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t4 - t3, t5 - t4, t6 - t5, t7 - t6)))
