import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from voxelized_geometry_tools_amd import capi
ctx = capi.Context(0); ctx.set_stream(None)
import os
shapes = [(2048,2048,128),(2048,2048,256)] if os.environ.get("VGT_HULL_W") else [(1024,1024,1024),(2048,2048,128),(2048,2048,256),(2048,2048,512),(2048,1024,256),(1024,2048,256)]
for shape in shapes:
    occ = bench.device_occupancy(torch, shape, "spheres", 42, torch.device("cuda",0), 0, (2048,2048,1024) if shape[0]==2048 and shape[1]==2048 else shape)
    sdf = torch.empty(shape, dtype=torch.float32, device="cuda")
    nb = capi.sdf_workspace_bytes(shape); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    mm = torch.zeros(2, device="cuda"); ms = np.zeros(3, dtype=np.float32); acc = np.zeros(3)
    for i in range(4):
        ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nb, mm.data_ptr(), kernel_ms=ms)
        if i: acc += ms
    acc /= 3; vox = np.prod(shape)
    print(shape, "ms", acc.round(2), "total %.2f" % acc.sum(), "Gvox/s %.1f" % (vox/acc.sum()/1e6))
    del occ, sdf, ws
