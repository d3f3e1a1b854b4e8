#!/usr/bin/env python3
"""Regenerate every committed tw_h?*_asm.inc / *_clobbers.inc under timewarp_amd/csrc from the generators - the same
invocation list tests/test_host_logic.py::test_generated_asm_includes_are_current checks the committed text against."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "tests", "test_host_logic.py")).read()
body = src[src.index("def test_generated_asm_includes_are_current"):]
body = body[body.index("for args in ("):body.index("subprocess.run(")]
calls = [eval(m) for m in re.findall(r"\[\"tools/gen_h3[^\]]*\]", body)]
env = {k: v for k, v in os.environ.items() if not k.endswith("_EXPERIMENT")}
for args in calls:
    subprocess.run([sys.executable] + args, cwd=root, check=True, env=env, stdout=subprocess.DEVNULL)
print(f"{len(calls)} generator runs")
