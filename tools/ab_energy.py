import os, sys, json, subprocess
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
# A/B on one box: tw_mh_iteration's energy kernel on the side stream (bit 23) against the caller's stream (bit 22);
# TW_AB_ARGS="--config dense" etc. passes bench arguments through.
for rep in range(2):
    for flags in (8388608, 4194304):
        env = dict(os.environ, TW_AB_DEBUG_FLAGS=str(flags))
        out = subprocess.run([sys.executable, "-c", "import os,ctypes,runpy,sys; from timewarp_amd import _lib; _lib.load().tw_debug_set_flags(int(os.environ['TW_AB_DEBUG_FLAGS'])); sys.argv=['bench.py','--no-cpu-baseline','--steps','40']+os.environ.get('TW_AB_ARGS','').split(); runpy.run_path('bench.py', run_name='__main__')"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print("inline" if flags == 4194304 else "side  ", round(d["value"], 2), round(d["ms_per_step"], 4), round(d["roofline"]["avg_launch_ms"], 4), flush=True)
