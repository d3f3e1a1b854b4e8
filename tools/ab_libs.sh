#!/bin/bash
# A/B on ONE box of timing-only library variants (timewarp_amd/lib/ab/libtimewarp_hip_<name>.so, tools/build_h3_experiment.sh):
# reverse + forward pass of the headline shape (tools/time_flow.py --paths 3, no range-guard read-back), interleaved rounds.
# usage: tools/ab_libs.sh <name> [<name> ...]
cd "$(dirname "$0")/.."
one() { python tools/time_flow.py --paths 3 --iters 30 2>/dev/null | grep "ms" | awk -v l="$1" '{s+=$3} END {printf "%-12s reverse+forward %.3f ms\n", l, s}'; }
for rep in 1 2 3; do
  one "in-tree"
  for n in "$@"; do TW_HIP_LIB=$PWD/timewarp_amd/lib/ab/libtimewarp_hip_$n.so one "$n"; done
done
