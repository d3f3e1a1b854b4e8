"""Where an MH iteration's time goes between kernels: from a `rocprofv3 --kernel-trace` CSV of `bench.py`, the last full
iterations' kernels in launch order with their durations and the idle gap in front of each (same stream; the energy
kernel runs on a side stream and shows up as an overlap)."""
import csv, glob, sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: the last 40 % of the trace, cut at mh_begin kernels
names = [r["Kernel_Name"] for r in rows]
want = sys.argv[2] if len(sys.argv) > 2 else "netblock_h3_kernel"   # iterations of this kernel family
begins = [i for i, n in enumerate(names) if "mh_begin_kernel" in n]
begins = [b for k, b in enumerate(begins[:-1]) if any(want in n for n in names[b:begins[k + 1]])]
if len(begins) < 6:
    sys.exit("no MH iterations in the trace")
lo, hi = begins[-5], begins[-1]
it = rows[lo:hi]
n_it = 4
tot = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / n_it
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it) / n_it
print(f"{n_it} iterations: {tot / 1e3:.1f} us each, kernels {busy / 1e3:.1f} us (side-stream overlap counted twice), "
      f"{len(it) // n_it} kernels per iteration")
one = rows[begins[-2]:begins[-1]]
prev_end = None
for r in one:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"  gap {gap:7.1f} us   {((e - s) / 1e3):8.1f} us  {r['Kernel_Name'][:100]}")
    prev_end = max(prev_end or 0, e)
