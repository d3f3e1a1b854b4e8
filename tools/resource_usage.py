#!/usr/bin/env python3
"""Compile csrc/tw_netblock_h3.hip (or the file given) with -Rpass-analysis=kernel-resource-usage and print one line per
kernel: demangled name, VGPRs, AGPRs, SGPRs, scratch bytes per lane, and every compiler warning.  No GPU needed.

    python tools/resource_usage.py [source.hip] > profiles/rNN_resource_usage.txt
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "timewarp_amd", "csrc", "tw_netblock_h3.hip")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", src,
           "-o", "/dev/null"]
    res = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stderr)
        sys.exit(res.returncode)
    rows, cur = [], None
    for line in res.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    names = [r["name"] for r in rows]
    dem = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True).stdout.splitlines()
    print(f"# {' '.join(cmd[:-2])}")
    print(f"{'VGPRs':>6} {'AGPRs':>6} {'SGPRs':>6} {'scratch B/lane':>15}  kernel")
    for r, d in zip(rows, dem):
        d = re.sub(r"\(.*\)$", "", d).replace("void ", "")
        print(f"{r.get('VGPRs', -1):>6} {r.get('AGPRs', -1):>6} {r.get('SGPRs', -1):>6} {r.get('ScratchSize', -1):>15}  {d}")
    warns = [l for l in res.stderr.splitlines() if "warning:" in l and "is not a recognized feature" not in l]
    print(f"# warnings: {len(warns)}")
    for w in warns:
        print("#   " + w)


if __name__ == "__main__":
    main()
