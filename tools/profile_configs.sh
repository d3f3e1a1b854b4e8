# BASELINE configs[3] / [4] on the GPU box: the bench line of each (live HIP-event roofline) and a rocprofv3 --kernel-trace
# --stats pass of the same command, whose average duration for the dominant kernel must agree (profiles/r04_*).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for c in 4aa dense; do
  python $R/bench.py --config $c --steps 30 --warmup 3 > $O/r04_bench_$c.json 2> $O/r04_bench_$c.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_$c -- python $R/bench.py --config $c --steps 30 --warmup 3 > $O/r04_bench_${c}_under_rocprof.json 2> $O/prof_r04_$c.err
  python $R/tools/summarize_profiles.py --stats $O/prof_r04_$c $O/r04_bench_${c}_kernel_stats.csv | head -6
  python - <<PY
import json
d = json.load(open("$O/r04_bench_$c.json"))
r = d["roofline"]
print("$c", d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["avg_launch_ms"], r["algorithmic_tflop_per_iteration"])
PY
done
