import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from voxelized_geometry_tools_amd import capi, synthetic
n=256; counts=(n,n,n); vs=np.float32(5.12/n); ivs=np.float32(1.0)/vs; sizes=[np.float32(c)*vs for c in counts]
pts = synthetic.raycast_cloud(1_000_000, seed=42)
# order by direction: octahedral-ish bins via azimuth/elevation
d = pts / np.maximum(np.linalg.norm(pts, axis=1, keepdims=True), 1e-9)
az = np.arctan2(d[:,1], d[:,0]); el = np.arcsin(np.clip(d[:,2], -1, 1))
def order(bins):
    a = np.floor((az + np.pi) / (2*np.pi) * bins).astype(np.int64).clip(0, bins-1)
    e = np.floor((el + np.pi/2) / np.pi * bins).astype(np.int64).clip(0, bins-1)
    # serpentine / tile order: morton of (a, e)
    def part(x):
        x = x & 0xffff; x = (x | (x << 8)) & 0x00ff00ff; x = (x | (x << 4)) & 0x0f0f0f0f; x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555; return x
    return np.argsort(part(a) | (part(e) << 1), kind="stable")
ctx = capi.Context(0); ctx.set_stream(None)
xf = synthetic.translation_xform(2.56,2.56,2.56).astype(np.float32)
for name, idx in (("random", None), ("sorted64", order(64)), ("sorted256", order(256)), ("sorted1024", order(1024))):
    p = pts if idx is None else pts[idx]
    pd = torch.from_numpy(np.ascontiguousarray(p)).cuda()
    grids = ctx.tracking_grids(n**3, 1)
    for _ in range(3): grids.raycast_f32_dev(0, pd.data_ptr(), len(p), 3.0, xf, vs, ivs, sizes, counts)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): grids.raycast_f32_dev(0, pd.data_ptr(), len(p), 3.0, xf, vs, ivs, sizes, counts)
    torch.cuda.synchronize(); ms=(time.perf_counter()-t0)/10*1e3
    print(name, "%.3f ms" % ms, "%.0f Mpoints/s" % (len(p)/ms/1e3))
    grids.close()
