#!/bin/bash
# bench_raycast.py against every diagnostic build of the library under diag_*/ (VGT_HIP_LIB override): cloud A / B raycast
# time and whether the counts still equal the oracle's.
for d in "" $(ls -d diag_* 2>/dev/null); do
  if [ -z "$d" ]; then lib=voxelized_geometry_tools_amd/libvgt_hip.so; name=product; else lib=$d/libvgt_diag.so; name=$d; fi
  VGT_HIP_LIB=$PWD/$lib timeout 200 python bench_raycast.py "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', {k:(v['raycast_ms'], v.get('counts_bit_exact')) for k,v in d['results'].items()})"
done
