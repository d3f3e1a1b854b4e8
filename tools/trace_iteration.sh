#!/bin/bash
# One MH iteration of the headline bench as a kernel timeline: start / end of every kernel between two accept kernels.
# Usage (GPU box): tools/trace_iteration.sh [bench args] -> gpurun_out/iter_trace.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/iter_trace && mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/iter_trace -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/iter_trace_bench.log 2>&1
python - <<'PY' > gpurun_out/iter_trace.txt
import csv, glob
f = glob.glob('gpurun_out/iter_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
acc = [i for i, r in enumerate(rows) if 'mh_accept_full' in r['Kernel_Name'] or 'mhc_accept' in r['Kernel_Name']]
# the 10th and 11th iteration of the headline leg (the alt paths come later in the run): with --warmup 3 and a read-back every 8
# iterations the host's flush - synchronisation + bookkeeping, GPU idle - sits between them
a, b = acc[9], acc[11]
t0 = int(rows[a]['End_Timestamp'])
prev = t0
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:90]}")
    prev = max(prev, e)
    busy += e - s
print(f"iteration: {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us end to end, {busy / 1e3:.1f} us of kernel time (all queues)")
PY
tail -3 gpurun_out/iter_trace_bench.log
