#!/usr/bin/env python3
"""Scan the device assembly of every kernel for an MFMA issued under an EXEC-mask predicate without a branch.

MFMA ignores EXEC.  When hipcc believes a condition is divergent (anything derived from threadIdx that was not passed
through readfirstlane) it predicates the guarded block with s_and_saveexec, and for a short block it drops the
s_cbranch_execz skip -- the "skipped" MFMA then runs anyway.  (Found the hard way in skinny.hip's k-tail.)

    python tools/scan_mfma_exec.py            # compiles eqxvision_amd/csrc/*.hip to /tmp and scans
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(path):
    lines = open(path).read().split("\n")
    hits, kern = [], None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        if "saveexec" not in l:
            continue
        for j in range(i + 1, min(i + 60, len(lines))):
            t = lines[j]
            if "s_cbranch_exec" in t or re.search(r"s_(or|mov|xor)_b64 exec", t):
                break
            if "v_mfma" in t:
                hits.append((kern, j + 1))
                break
    return hits


def main():
    out = tempfile.mkdtemp(prefix="mfma_exec_")
    procs = []
    for src in sorted(glob.glob(os.path.join(ROOT, "eqxvision_amd/csrc/*.hip"))):
        s = os.path.join(out, os.path.basename(src)[:-4] + ".s")
        procs.append((s, subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                                            "--cuda-device-only", src, "-o", s], stderr=subprocess.DEVNULL)))
    bad = 0
    for s, p in procs:
        if p.wait() != 0:
            print("compile failed:", s)
            bad += 1
            continue
        for kern, line in scan(s):
            print(f"{os.path.basename(s)}:{line}: MFMA under EXEC predicate without branch in {kern[:80]}")
            bad += 1
    print("suspicious:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
