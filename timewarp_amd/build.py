"""Build libtimewarp_hip.so for gfx950 with hipcc (in-tree, so the .so travels to the GPU box).

    python -m timewarp_amd.build            # rebuild if any source is newer than the library
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.environ.get("TW_HIP_LIB") or os.path.join(LIB_DIR, "libtimewarp_hip.so")  # TW_HIP_LIB: use a prebuilt library
SOURCES = ["tw_kernels.hip", "tw_netblock.hip", "tw_netblock_h3.hip", "tw_energy.hip", "tw_api.hip"]
HEADERS = [os.path.join(CSRC, "tw_common.h"), os.path.join(HERE, "..", "include", "timewarp_hip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


# Per-file extra flags (none needed at present; building the split-fp16 kernel without packed-fp32 VALU
# ops, which stall behind the matrix pipe in tools/probe/mfma_valu_overlap.hip, was measured: no gain).
EXTRA_FLAGS = {}


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 and link the C-ABI shared library."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libtimewarp_hip.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(src, []) + [
            "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = "\n".join(l for l in res.stderr.splitlines() if "is not a recognized feature for this target" not in l)
        if err.strip():
            sys.stderr.write(err + "\n")
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src} (exit {res.returncode})")
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
