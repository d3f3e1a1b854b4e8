set -x
python tools/stress_h3.py > gpurun_out/r02_stress.txt 2>&1; tail -5 gpurun_out/r02_stress.txt
TW_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 12 --warmup 2 > gpurun_out/r02_bench_gloo2_plumbing.json 2> gpurun_out/r02_bench_gloo2.err; tail -c 1500 gpurun_out/r02_bench_gloo2_plumbing.json; tail -3 gpurun_out/r02_bench_gloo2.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > gpurun_out/r02_bench4.json 2> gpurun_out/r02_bench4.err; python -c "
import json; d=json.load(open('gpurun_out/r02_bench4.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step']-16*d['roofline']['avg_launch_ms']); print(d['alt_path']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['thread_sweep_proposals_per_s'], d['cpu_baseline']['s_per_iteration'])"
