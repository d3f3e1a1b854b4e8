# Round evidence on the GPU box: kernel-trace stats of the bench command, FETCH/WRITE traffic (separate --pmc passes, never
# combined with other trace domains), SQ counters of the dominant kernel.  Summaries land in gpurun_out/ as r03_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_stats -- python $R/bench.py --steps 100 --warmup 3 --no-cpu-baseline > $O/r03_prof_bench_steps100.json 2> $O/prof_r03_stats.err
python $R/tools/summarize_profiles.py --stats $O/prof_r03_stats $O/r03_bench_h3_steps100_kernel_stats.csv | head -12
bash $R/tools/pmc_traffic.sh
python $R/tools/summarize_profiles.py --traffic $O/pmc_traffic_fetch $O/pmc_traffic_write $O/r03_pmc_traffic.json
bash $R/tools/pmc_h3.sh
python $R/tools/summarize_profiles.py --sq $O/pmc_h3_1 $O/pmc_h3_2 $O/pmc_h3_3 $O/pmc_h3_4 $O/r03_h3_sq_counters.md
python $R/tools/profile_h3_sections.py > $O/r03_h3_sections.txt 2>&1; tail -20 $O/r03_h3_sections.txt
