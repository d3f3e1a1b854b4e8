#!/bin/bash
# A/B on ONE box: the fast mode's 48-token build on the six-slot ring (the in-tree library) against the five-slot arrangement
# (timewarp_amd/lib/ab/libtimewarp_hip_ring5.so = tw_netblock_h3.hip compiled with -DTW_H1_RING5, see DESIGN_LOG 5.7).
# Build the five-slot variant first (in this container; the .so travels with the gpurun snapshot, lib/ab/ is git-ignored):
#   mkdir -p timewarp_amd/lib/ab && cd timewarp_amd/lib && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTW_H1_RING5 \
#     -c ../csrc/tw_netblock_h3.hip -o ab/h3_ring5.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libtimewarp_hip_ring5.so \
#     tw_api.o tw_energy.o tw_kernels.o tw_md.o tw_mh_step.o tw_netblock.o tw_netblock_dense.o ab/h3_ring5.o
cd "$(dirname "$0")/.."
[ -f timewarp_amd/lib/ab/libtimewarp_hip_ring5.so ] || { echo "build timewarp_amd/lib/ab/libtimewarp_hip_ring5.so first (see the header of this script)"; exit 1; }
one() { python bench.py --path h1 --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'accepted/s', round(d['roofline']['avg_launch_ms']*1e3,1), 'us/launch', round(d['roofline']['frac'],4))"; }
for rep in 1 2 3; do
  TW_HIP_LIB=$PWD/timewarp_amd/lib/ab/libtimewarp_hip_ring5.so one "ring 5:"
  one "ring 6:"
done
