# Per-kernel shares of a reverse pass on the per-op split path: rocprofv3 --kernel-trace --stats of tools/time_per_op.py
#   [PER_OP_ARGS=--forward] bash tools/profile_per_op.sh 256x256 [more sizes]     -> gpurun_out/per_op_<size>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for s in "$@"; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_per_op_$s -- python $R/tools/time_per_op.py --path=5 $PER_OP_ARGS $s > $O/per_op_$s.txt 2> $O/per_op_$s.err
  python $R/tools/summarize_profiles.py --stats $O/prof_per_op_$s $O/per_op_${s}_kernel_stats.csv | grep -v "h3_pack\|h3_copy" | head -14
  cat $O/per_op_$s.txt
  rm -rf $O/prof_per_op_$s
done
