"""A/B timing of generator variants of the split-fp16 kernel's asm sections, on the GPU box.

    python tools/ab_h3.py base auxrot nobarrier ...        # variant = comma list for H3_FFN_EXPERIMENT,
                                                           # 'attn:<flags>' for H3_ATTN_EXPERIMENT, 'base' = none

For every variant: regenerate timewarp_amd/csrc/tw_h3_*_asm.inc with the experiment flags, rebuild the library, run
tools/time_flow.py --paths 3 in a fresh process; rounds are interleaved (variant order repeated `--rounds` times) so box
drift shows.  The committed includes are regenerated at the end.  Results that delete work (noepi, nobarrier) are timing
experiments only."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regen(ffn_flags, attn_flags):
    env = dict(os.environ, H3_FFN_EXPERIMENT=ffn_flags, H3_ATTN_EXPERIMENT=attn_flags)
    for shape in ("ffn", "in", "out"):
        subprocess.run([sys.executable, "tools/gen_h3_ffn_asm.py", f"--shape={shape}"], cwd=ROOT, env=env, check=True,
                       stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, "tools/gen_h3_attn_asm.py"], cwd=ROOT, env=env, check=True, stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, "tools/gen_h3_attn_asm.py", "--mode=windowed"], cwd=ROOT, env=env, check=True,
                   stdout=subprocess.DEVNULL)
    # the encoder-stack statement (what the kernel runs by default) embeds both blocks: regenerate it with the same flags
    subprocess.run([sys.executable, "tools/gen_h3_enc_asm.py"], cwd=ROOT, env=env, check=True, stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, "tools/gen_h3_enc_asm.py", "--mode=windowed"], cwd=ROOT, env=env, check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, "-m", "timewarp_amd.build"], cwd=ROOT, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)


def time_variant(iters):
    out = subprocess.run([sys.executable, "tools/time_flow.py", "--paths", "3", "--iters", str(iters)], cwd=ROOT,
                         capture_output=True, text=True)
    ms = [float(m) for m in re.findall(r": ([0-9.]+) ms", out.stdout)]
    if len(ms) != 2:
        print(out.stdout[-2000:], out.stderr[-2000:])
        return None
    return ms


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = 2
    iters = 20
    for a in sys.argv[1:]:
        if a.startswith("--rounds="):
            rounds = int(a.split("=")[1])
        if a.startswith("--iters="):
            iters = int(a.split("=")[1])
    variants = args or ["base"]
    results = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            ffn = ",".join(f for f in v.split("+") if not f.startswith("attn:") and f != "base")
            attn = ",".join(f[5:] for f in v.split("+") if f.startswith("attn:"))
            regen(ffn, attn)
            ms = time_variant(iters)
            results[v].append(ms)
            print(f"round {r} {v:30s} reverse/forward pass ms: {ms}", flush=True)
    regen("", "")
    print("--- summary (min over rounds of reverse+forward, ms)")
    for v in variants:
        ok = [sum(m) for m in results[v] if m]
        print(f"{v:30s} {min(ok):.3f}" if ok else f"{v:30s} failed")


if __name__ == "__main__":
    main()
