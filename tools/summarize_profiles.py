"""Summaries of the rocprofv3 outputs that tools/profile_round.sh leaves under gpurun_out/ (run on the GPU box):
  --traffic  DIR_FETCH DIR_WRITE OUT.json [MERGE.json]  FETCH_SIZE / WRITE_SIZE per launch of the net-block kernels (gfx950 correction)
  --sq       DIR1 DIR2 ... OUT.md           SQ counters of netblock_h3 per wave
  --sqk      PATTERN WAVES DIR1 ... OUT.md  the same for the kernels whose name contains PATTERN, WAVES waves per launch
  --stats    DIR OUT.csv                    copy of the kernel-stats CSV of a --kernel-trace --stats run"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def counters(d):
    rows = []
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def traffic(d_fetch, d_write, out, merge=None, bench_args=""):
    res = json.load(open(merge)) if merge and os.path.exists(merge) else {}   # add this run's kernels to an earlier summary
    res.update({"command": "rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python bench.py --steps 3 --warmup 1 "
                      "--no-cpu-baseline (one pass per counter: FETCH_SIZE, WRITE_SIZE)",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (rocprofv3); bytes = value * 1024; FETCH_SIZE doubled for the "
                    "16 B/lane coalesced streams of these kernels (MI355X_MICROARCH.md section HBM)"})
    fresh = set()
    per = defaultdict(lambda: defaultdict(list))
    for name, d in (("FETCH_SIZE", d_fetch), ("WRITE_SIZE", d_write)):
        for r in counters(d):
            if r["Counter_Name"] == name:
                per[r["Kernel_Name"]][name].append(float(r["Counter_Value"]))
    for k, v in per.items():
        if "netblock" not in k:
            continue
        # the split-fp16 family by instantiation: <NT, ASM, DENSE, WIDE, RFF, ENC, H1> - the fast mode (H1) streams half the bytes
        args = [a.strip() for a in k[k.index("<") + 1:k.index(">")].split(",")] if "<" in k else []
        if "h3" in k and len(args) >= 7:
            base = "h1" if args[6] == "true" else "h3"
            key = f"netblock_{base}_dense_kernel" if args[2] == "true" else f"netblock_{base}_wide_kernel" if args[3] == "true" \
                else f"netblock_{base}_n4_kernel" if args[0] == "4" else f"netblock_{base}_kernel"
        elif "h3" in k:
            key = "netblock_h3_n4_kernel" if args[:1] == ["4"] else "netblock_h3_kernel"
        else:
            key = "netblock_dense_kernel" if "dense" in k else "netblock_kernel"
        f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
        w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
        # r05: one record per (instantiation family, bench configuration) - bench.py takes a figure only where both match
        key += f" [{bench_args.strip()}]" if bench_args.strip() else ""
        if key in fresh and res[key]["dispatches"] >= len(v["FETCH_SIZE"]):
            continue  # several instantiations of one kernel family in the run: keep the one the timed region launches
        fresh.add(key)
        res[key] = {"kernel": k, "bench_args": bench_args.strip(), "dispatches": len(v["FETCH_SIZE"]), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                    "traffic_bytes_per_launch_corrected": (2 * f + w) * 1024,
                    "traffic_bytes_per_launch_uncorrected": (f + w) * 1024,
                    "note": "fabric-side L2 requests (Infinity-Cache hits are counted): the 8 XCD L2s each stream one net's "
                            "weight stages out of the Infinity Cache; not HBM-limited"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k.startswith("netblock")}, indent=1))


def sq(dirs, out, pattern="netblock_h3", n_waves=None):
    acc = defaultdict(list)
    for d in dirs:
        for r in counters(d):
            if pattern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # waves per launch: 2 nets x ceil(rows / molecules per workgroup) workgroups x 4 waves - 1000 for the 1000-proposal
    # alanine-dipeptide launches (250 workgroups), 2048 for NNQQ x 512 proposals on the wide layout (2 molecules per workgroup)
    # (paired layout, 100 atoms x 512 proposals: 256 workgroups per net as well)
    waves = float(n_waves) if n_waves else (2048.0 if ("4aa" in dirs[0] or "nnqq" in dirs[0] or "paired" in dirs[0]) else 1000.0)
    names = sorted({r["Kernel_Name"] for d in dirs for r in counters(d) if pattern in r["Kernel_Name"]})
    lines = [f"# SQ counters, {', '.join('`' + n.split('(')[0].replace('void ', '') + '`' for n in names)}", "",
             f"`bash tools/pmc_h3.sh` on the GPU box ({os.path.basename(dirs[0])[:-2]}): four `rocprofv3 --kernel-trace --pmc <4 counters> "
             f"--kernel-include-regex {pattern}` passes (command in tools/pmc_h3.sh). "
             f"Averages per launch divided by the {waves:.0f} waves of a launch; SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles, "
             "SQ_VALU_MFMA_BUSY_CYCLES counts clocks.", "", "| counter | per wave |", "|---|---|"]
    vals = {k: sum(v) / len(v) / waves for k, v in acc.items()}
    for k in sorted(vals):
        lines.append(f"| {k} | {vals[k] / 1e3:.1f} k |")
    if "SQ_WAVE_CYCLES" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        lines += ["", f"matrix pipe busy: {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 4 / vals['SQ_WAVE_CYCLES'] * 100:.1f} % of the wave cycles; "
                      f"parked at waits / barriers: {vals.get('SQ_WAIT_ANY', 0) / vals['SQ_WAVE_CYCLES'] * 100:.1f} %; "
                      f"non-MFMA VALU instructions per MFMA: "
                      f"{(vals.get('SQ_INSTS_VALU', 0) - vals.get('SQ_INSTS_MFMA', 0)) / max(vals.get('SQ_INSTS_MFMA', 1), 1):.2f}"]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-3:]))


def stats(d, out):
    f = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))
    shutil.copy(f[0], out)
    print(open(out).read()[:1500])


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "--traffic":
        # optional 4th argument: an earlier summary to merge into; 5th: the bench.py arguments the passes ran with ("" = headline)
        traffic(*sys.argv[2:6], **({"bench_args": sys.argv[6]} if len(sys.argv) > 6 else {}))
    elif mode == "--sq":
        sq(sys.argv[2:-1], sys.argv[-1])
    elif mode == "--sqk":
        sq(sys.argv[4:-1], sys.argv[-1], sys.argv[2], sys.argv[3])
    elif mode == "--stats":
        stats(sys.argv[2], sys.argv[3])
