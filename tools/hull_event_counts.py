import ctypes, os, sys
sys.path.insert(0, "/root/repo")
os.environ["VGT_HIP_LIB"] = "/root/repo/voxelized_geometry_tools_amd/libvgt_hip_stats2.so"
import numpy as np, torch, bench
from voxelized_geometry_tools_amd import capi
size = 512; dist = sys.argv[1] if len(sys.argv) > 1 else "spheres"
shape = (size,)*3
occ = bench.device_occupancy(torch, shape, dist, 42, torch.device("cuda", 0))
sdf = torch.empty(shape, dtype=torch.float32, device="cuda")
nb = capi.sdf_workspace_bytes(shape); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
ctx = capi.Context(0); ctx.set_stream(None)
lib = capi.load(); out = (ctypes.c_ulonglong * 32)()
lib.vgt_hip_debug_hull_stats(out, 1)
ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nb); torch.cuda.synchronize()
lib.vgt_hip_debug_hull_stats(out, 1)
vox = float(np.prod(shape)); bands = vox / 32
for base, name in ((0, "Y"), (16, "X")):
    v = [out[base+i] for i in range(16)]
    print(name, "per band: finite %.2f cand %.2f predicates %.2f pops %.2f merge_kills %.2f survivors(owners) %.2f lane_iters %.2f wave_max_iters %.2f (waves=%d)" % (
        v[11]/bands, v[13]/bands, v[7]/bands, v[8]/bands, v[9]/bands, v[10]/bands, v[14]/bands, v[15]/(bands/64), bands/64))
