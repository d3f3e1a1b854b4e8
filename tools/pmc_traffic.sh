# HBM/fabric traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (never combined with sys/hip/hsa traces).  Output under gpurun_out/pmc_traffic_{fetch,write}/.
#   bash tools/pmc_traffic.sh [bench.py arguments, e.g. --config 4aa]   (default: the headline configuration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  rm -rf $R/gpurun_out/pmc_traffic_$n
  timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "netblock" --output-format csv -d $R/gpurun_out/pmc_traffic_$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_traffic_$n.log 2>&1
  echo "$c rc=$?"
done
