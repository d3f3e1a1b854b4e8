#!/bin/bash
# One MH iteration on the 691-atom test protein (tools/time_protein_iteration.py, 16 proposals) as a kernel timeline summary: per queue
# (the caller's stream, the flow's side stream for the second net of a coupling layer, the energy kernel's side stream) the busy time,
# per kernel name the calls and the time, and the iteration end to end.  Usage (GPU box): tools/trace_protein_iteration.sh -> gpurun_out/protein_iter_trace.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/protein_trace && mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/protein_trace -- python tools/time_protein_iteration.py 16 > gpurun_out/protein_trace.log 2>&1
python - <<'PY' > gpurun_out/protein_iter_trace.txt
import csv, glob
from collections import defaultdict
f = glob.glob('gpurun_out/protein_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
acc = [i for i, r in enumerate(rows) if 'mh_accept_full' in r['Kernel_Name'] or 'mhc_accept' in r['Kernel_Name']]
a, b = acc[4], acc[5]   # the third timed iteration
t0, t1 = int(rows[a]['End_Timestamp']), int(rows[b]['End_Timestamp'])
per_q, per_k = defaultdict(int), defaultdict(lambda: [0, 0])
for r in rows[a + 1:b + 1]:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    per_q[r.get('Queue_Id', '?')] += d
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
    per_k[name][0] += 1
    per_k[name][1] += d
print(f"one MH iteration on 1hgv (691 atoms x 16 proposals): {(t1 - t0) / 1e3:.0f} us end to end, {sum(per_q.values()) / 1e3:.0f} us of kernel time over {len(per_q)} queues, {b - a} launches")
for q, d in sorted(per_q.items(), key=lambda x: -x[1]):
    print(f"  queue {q}: {d / 1e3:8.0f} us busy")
for k, (n, d) in sorted(per_k.items(), key=lambda x: -x[1][1])[:16]:
    print(f"  {d / 1e3:8.0f} us  {n:4d} x  {k}")
PY
cat gpurun_out/protein_iter_trace.txt
