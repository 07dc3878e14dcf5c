"""Build libks265hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo snapshot)."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libks265hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-unused-function"]
FLAGS += os.environ.get("KS265_EXTRA_FLAGS", "").split()          # tuning experiments only (e.g. -DKS_SUBPEL_NT=64)


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_sha() -> str:
    """identity of the kernel sources (every .hip / .h under csrc + the C-ABI header): profiles/*.json carry the value they were measured with, bench.py compares it with
    the running code's and flags stale counters instead of dividing old counts by new durations"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "ks265_hip.h")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "ks265_hip.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for s in sources():
        o = os.path.join(CSRC, "build", os.path.basename(s) + ".o")
        objs.append(o)
        procs.append((s, subprocess.Popen([HIPCC, *FLAGS, "-c", s, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
