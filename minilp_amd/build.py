"""Builds minilp_amd/libminilp_hip.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libminilp_hip.so")
SOURCES = ["kernels.hip", "engine.hip", "capi.hip", "mps.cpp", "util.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "minilp_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(src):
    """Files an object depends on: its source, every header, and (kernels.hip only) the .inc files it includes."""
    inc = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or (src == "kernels.hip" and f.endswith(".inc"))]
    return [os.path.join(CSRC, src), os.path.join(HERE, "..", "include", "minilp_hip.h")] + inc


def build(force=False, verbose=True):
    if not force and not needs_build():
        return SO
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src)):
            continue  # (per-object: a change to engine.hip does not recompile the kernels)
        cmd = [hipcc()] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:  # the translation units compile side by side
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
