"""Scan hipcc's gfx950 assembly for epilogues whose global / buffer stores are serialised by `s_waitcnt vmcnt(0|1)`:
gfx9's vmcnt counts loads and stores in issue order, so a conservative wait that sits between two stores costs one
store round trip (round 5: the igemm8 epilogue had 16 of them per wave and tile).  usage: scan_store_waits.py FILE.s ..."""
import re, sys, subprocess

def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()[:110]
    except Exception:
        return n

for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s", l)]
    starts.append((len(lines), "end"))
    for k in range(len(starts) - 1):
        a, name = starts[k]
        body = lines[a:starts[k + 1][0]]
        nst = 0; serial = 0; since = 0
        for l in body:
            t = l.strip()
            if re.match(r"(global|buffer|flat)_store", t):
                nst += 1; since += 1
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if m and int(m.group(1)) <= 1 and since > 0:
                serial += 1; since = 0
        if nst >= 2 and serial >= (int(__import__("os").environ.get("MINSER","3"))):
            print(f"{path.split('/')[-1]:24s} stores {nst:4d}  serialising waits {serial:3d}  {demangle(name)}")
