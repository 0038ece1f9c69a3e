import ctypes, os, sys
sys.path.insert(0, "/root/repo"); os.environ["VGT_HIP_LIB"]="/root/repo/voxelized_geometry_tools_amd/libvgt_hip_stats.so"
import numpy as np, torch, bench
from voxelized_geometry_tools_amd import capi
size=int(sys.argv[1]); dist=sys.argv[2]
shape=(size,)*3; dev=torch.device("cuda",0)
occ=bench.device_occupancy(torch,shape,dist,42,dev); sdf=torch.empty(shape,dtype=torch.float32,device=dev)
nb=capi.sdf_workspace_bytes(shape); ws=torch.empty(nb,dtype=torch.uint8,device=dev)
ctx=capi.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); lib=capi.load()
out=(ctypes.c_ulonglong*32)(); lib.vgt_hip_debug_hull_stats(out,1)
ctx.sdf_dev(occ.data_ptr(),shape,0.01,sdf.data_ptr(),ws.data_ptr(),nb); torch.cuda.synchronize()
lib.vgt_hip_debug_hull_stats(out,1)
vox=float(np.prod(shape)); waves=vox/32/64
for base,name in ((0,"Y"),(16,"X")):
    v=[out[base+i] for i in range(16)]
    print(name,dist,"pred/vox %.3f pops/vox %.3f kills/vox %.4f surv/vox %.4f finite/vox %.3f cand/vox %.3f | lane iters/band %.2f wave-max iters/band %.2f"%(v[7]/vox,v[8]/vox,v[9]/vox,v[10]/vox,v[11]/vox,v[13]/vox,v[14]/(vox/32),v[15]/waves))
