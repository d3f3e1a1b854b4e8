# Round-6 evidence on the GPU box.  Per bench configuration (ad = the default command, whose line also carries other_configs and the CPU
# baseline; 4aa; dense): the bench line, a rocprofv3 --kernel-trace --stats pass of the same command, FETCH / WRITE traffic in
# separate --pmc passes.  Lock-step chains (--chains 32 / 8): the bench line and, for 32, the kernel-stats CSV (no ATen kernel inside
# a step).  Summaries -> gpurun_out/r06_*.
#   bash tools/profile_round_r06.sh [configs...]      default: ad 4aa dense chains
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
CONFIGS=${@:-ad 4aa dense chains}
for c in $CONFIGS; do
  if [ $c = chains ]; then
    for n in 32 8; do
      timeout 600 python $R/bench.py --chains $n --steps 40 --warmup 5 > $O/r06_bench_chains_$n.json 2> $O/r06_bench_chains_$n.err
    done
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_chains -- python $R/bench.py --chains 32 --steps 40 --warmup 5 > $O/r06_bench_chains_32_under_rocprof.json 2> $O/prof_r06_chains.err
    python $R/tools/summarize_profiles.py --stats $O/prof_r06_chains $O/r06_bench_chains_32_kernel_stats.csv | head -12
    rm -rf $O/prof_r06_chains
    continue
  fi
  if [ $c = ad ]; then A=""; N="default"; else A="--config $c"; N=$c; fi
  timeout 900 python $R/bench.py $A --steps 30 --warmup 5 > $O/r06_bench_$N.json 2> $O/r06_bench_$N.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_$c -- python $R/bench.py $A --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r06_bench_${N}_under_rocprof.json 2> $O/prof_r06_$c.err
  python $R/tools/summarize_profiles.py --stats $O/prof_r06_$c $O/r06_bench_${N}_kernel_stats.csv | head -5
  bash $R/tools/pmc_traffic.sh $A --no-other-configs
  python $R/tools/summarize_profiles.py --traffic $O/pmc_traffic_fetch $O/pmc_traffic_write $O/r06_pmc_traffic.json $O/r06_pmc_traffic.json "$A" > /dev/null
  python - <<PY
import json
d = json.loads(open("$O/r06_bench_$N.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$c", round(d["value"], 1), round(d["ms_per_step"], 3), round(r["achieved"], 1), round(r["frac"], 4), round(r["avg_launch_ms"], 4), r["kernel"][:90], r.get("traffic"))
PY
  rm -rf $O/prof_r06_$c $O/pmc_traffic_fetch $O/pmc_traffic_write
done
