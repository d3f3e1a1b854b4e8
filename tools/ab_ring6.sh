#!/bin/bash
# A/B on ONE box: the fast mode's 48-token build on the six-slot ring (the in-tree library) against the five-slot arrangement
# (timewarp_amd/lib/ab/libtimewarp_hip_ring5.so = tw_netblock_h3.hip compiled with -DTW_H1_RING5, see DESIGN_LOG 5.7).
cd "$(dirname "$0")/.."
one() { python bench.py --path h1 --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'accepted/s', round(d['roofline']['avg_launch_ms']*1e3,1), 'us/launch', round(d['roofline']['frac'],4))"; }
for rep in 1 2 3; do
  TW_HIP_LIB=$PWD/timewarp_amd/lib/ab/libtimewarp_hip_ring5.so one "ring 5:"
  one "ring 6:"
done
