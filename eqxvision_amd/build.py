"""In-tree build of the HIP library: `hipcc --offload-arch=gfx950` on every `csrc/*.hip`, one object
per file (compiled in parallel), linked into `csrc/libeqxvision_amd.so`.  Cross-compiles without a GPU."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libeqxvision_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
PROF = bool(os.environ.get("EQV_PROF"))
if PROF:          # debug build: per-block / per-barrier time stamps in igemm2 / igemm8 (tools/phase_prof.py); its own library and
    FLAGS.append("-DMV_I8_PROF")      # object directory, loaded with EQV_LIB=<path> -- the product library is never a debug build
    LIB = os.path.join(CSRC, "libeqxvision_amd_prof.so")


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objdir = os.path.join(CSRC, "build_prof" if PROF else "build")
    os.makedirs(objdir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        if force or _newer(s, o) or newest_hdr > os.path.getmtime(o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{r.stderr[-4000:]}")
        return s

    if jobs:
        if verbose:
            print(f"[eqxvision_amd.build] compiling {len(jobs)} file(s) for gfx950", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
