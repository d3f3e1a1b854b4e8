set -x
for i in 1 2 3; do
python tools/time_flow.py --paths 3 --iters 30 --debug-flags 0 2>&1 | grep path
python tools/time_flow.py --paths 3 --iters 30 --debug-flags 32 2>&1 | grep path
done > gpurun_out/r02_ab_coupling.txt
cat gpurun_out/r02_ab_coupling.txt
python - <<'PY' > gpurun_out/r02_ab_step.txt 2>&1
import os, subprocess, json, sys
for rnd in range(2):
    for fused in ("1", "0"):
        env = dict(os.environ, TW_MH_FUSED=fused)
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "40"], env=env, capture_output=True, text=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("TW_MH_FUSED=" + fused, "value", round(d["value"], 2), "ms_per_step", round(d["ms_per_step"], 4), "launch", round(d["roofline"]["avg_launch_ms"], 4),
              "non-kernel", round(d["ms_per_step"] - 16 * d["roofline"]["avg_launch_ms"], 4))
PY
cat gpurun_out/r02_ab_step.txt
