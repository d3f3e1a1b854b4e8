# Round evidence on the GPU box: kernel-trace stats of the bench command, FETCH/WRITE traffic (separate --pmc passes, never
# combined with other trace domains), SQ counters of the dominant kernel.  Summaries land in gpurun_out/ as r04_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_stats -- python $R/bench.py --steps 100 --warmup 3 --no-cpu-baseline > $O/r04_prof_bench_steps100.json 2> $O/prof_r04_stats.err
python $R/tools/summarize_profiles.py --stats $O/prof_r04_stats $O/r04_bench_h3_steps100_kernel_stats.csv | head -12
bash $R/tools/pmc_traffic.sh
python $R/tools/summarize_profiles.py --traffic $O/pmc_traffic_fetch $O/pmc_traffic_write $O/r04_pmc_traffic.json
bash $R/tools/pmc_h3.sh
python $R/tools/summarize_profiles.py --sq $O/pmc_h3_1 $O/pmc_h3_2 $O/pmc_h3_3 $O/pmc_h3_4 $O/r04_h3_sq_counters.md
python $R/tools/profile_h3_sections.py > $O/r04_h3_sections.txt 2>&1; tail -20 $O/r04_h3_sections.txt
# r04: the fast mode (TW_PATH_FUSED_H1) and the kernels of BASELINE configs[3] / [4]: SQ counters of one flow pass / bench run each
bash $R/tools/pmc_h3.sh h1
python $R/tools/summarize_profiles.py --sq $O/pmc_h1_1 $O/pmc_h1_2 $O/pmc_h1_3 $O/pmc_h1_4 $O/r04_h1_sq_counters.md
bash $R/tools/pmc_h3.sh 4aa
python $R/tools/summarize_profiles.py --sq $O/pmc_4aa_1 $O/pmc_4aa_2 $O/pmc_4aa_3 $O/pmc_4aa_4 $O/r04_wide_sq_counters.md
bash $R/tools/pmc_h3.sh dense
python $R/tools/summarize_profiles.py --sq $O/pmc_dense_1 $O/pmc_dense_2 $O/pmc_dense_3 $O/pmc_dense_4 $O/r04_dense_sq_counters.md
python $R/tools/profile_h3_sections.py --h1 > $O/r04_h1_sections.txt 2>&1; tail -8 $O/r04_h1_sections.txt
# raw counter dumps are scratch: only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf $O/prof_r04_stats $O/pmc_*_[1-4] $O/pmc_traffic_fetch $O/pmc_traffic_write
