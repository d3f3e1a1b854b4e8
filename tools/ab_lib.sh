#!/bin/bash
# A/B on ONE box: bench.py with the in-tree library against timewarp_amd/lib/ab/libtimewarp_hip_<name>.so, interleaved.
# usage: tools/ab_lib.sh <name> [bench args]
cd "$(dirname "$0")/.."
name=$1; shift
one() { python bench.py --no-cpu-baseline --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$LABEL', round(d['value'],1), 'accepted/s', round(d['roofline']['avg_launch_ms']*1e3,1), 'us/launch', round(d['roofline']['frac'],4))"; }
for rep in 1 2 3; do
  LABEL="in-tree:" one "$@"
  LABEL="$name:" TW_HIP_LIB=$PWD/timewarp_amd/lib/ab/libtimewarp_hip_$name.so one "$@"
done
