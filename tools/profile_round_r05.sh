# Round-5 evidence on the GPU box, per bench configuration: the bench line (live HIP-event roofline), a rocprofv3 --kernel-trace
# --stats pass of the same command (its average duration for the dominant kernel must agree), FETCH / WRITE traffic in separate
# --pmc passes (never combined with other trace domains) and the SQ counters of the dominant kernel.  Summaries -> gpurun_out/r05_*.
#   bash tools/profile_round_r05.sh [configs...]      default: ad 4aa 4aa-nnqq dense
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
CONFIGS=${@:-ad 4aa 4aa-nnqq dense}
for c in $CONFIGS; do
  if [ $c = ad ]; then A=""; else A="--config $c"; fi
  timeout 600 python $R/bench.py $A --steps 30 --warmup 3 > $O/r05_bench_$c.json 2> $O/r05_bench_$c.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05_$c -- python $R/bench.py $A --steps 30 --warmup 3 --no-cpu-baseline > $O/r05_bench_${c}_under_rocprof.json 2> $O/prof_r05_$c.err
  python $R/tools/summarize_profiles.py --stats $O/prof_r05_$c $O/r05_bench_${c}_kernel_stats.csv | head -5
  bash $R/tools/pmc_traffic.sh $A
  python $R/tools/summarize_profiles.py --traffic $O/pmc_traffic_fetch $O/pmc_traffic_write $O/r05_pmc_traffic.json $O/r05_pmc_traffic.json "$A" > /dev/null
  case $c in ad) W=h3;; 4aa) W=4aa;; 4aa-nnqq) W=nnqq;; dense) W=dense;; esac
  bash $R/tools/pmc_h3.sh $W
  python $R/tools/summarize_profiles.py --sq $O/pmc_${W}_1 $O/pmc_${W}_2 $O/pmc_${W}_3 $O/pmc_${W}_4 $O/r05_${c}_sq_counters.md
  python - <<PY
import json
d = json.load(open("$O/r05_bench_$c.json"))
r = d["roofline"]
print("$c", round(d["value"], 1), round(d["ms_per_step"], 3), round(r["achieved"], 1), round(r["frac"], 4), round(r["avg_launch_ms"], 4), r["kernel"][:90], r.get("traffic"))
PY
  rm -rf $O/prof_r05_$c $O/pmc_${W}_[1-4] $O/pmc_traffic_fetch $O/pmc_traffic_write
done
python -c "
import json; d = json.load(open('$O/r05_pmc_traffic.json'))
for k, v in d.items():
    if isinstance(v, dict): print(k, v['kernel'][:100], 'FETCH KiB', round(v['FETCH_SIZE_KiB']), 'WRITE KiB', round(v['WRITE_SIZE_KiB']))
"
