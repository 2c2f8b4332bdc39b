"""In-tree build of the HIP library (libbbbp_hip.so) for gfx950 with hipcc.

No torch headers are involved: the library is a plain C-ABI shared object (include/bbbp_hip.h)
that Python binds with ctypes, so the same .so serves any host language.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbbbp_hip.so")
SOURCES = ["util.hip", "gemm.hip", "conv.hip", "conv_wino.hip", "conv_b3.hip", "conv_b3c1.hip", "rowops.hip", "engine.hip", "encoder.hip", "fold.hip", "attention.hip", "attention_b3.hip", "preprocess.hip", "mlp.hip", "head.hip", "forest.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-source flags.  conv_b3c1.hip: the pooling epilogue takes maxima of accumulator registers; in IEEE mode every such operand first gets a
# quieting `v_max_f32 x, x, x` (two extra vector instructions per maximum in a kernel whose vector issue slots are the budget).  Without the
# IEEE bit and with finite-math NaN rules the maxima are single instructions; NaN inputs are outside this path's contract.
EXTRA_FLAGS = {"conv_b3c1.hip": ["-mno-amdgpu-ieee", "-fno-honor-nans"]}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip under csrc/ for gfx950 and link libbbbp_hip.so.  Returns its path."""
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "bbbp_hip.h"))
    hdrs = [h for h in hdrs if os.path.exists(h)]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src, os.path.abspath(__file__)] + hdrs):
            jobs.append([HIPCC, *FLAGS, *EXTRA_FLAGS.get(s, []), "-I", os.path.join(os.path.dirname(HERE), "include"), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
