"""Per kernel of a hipcc .s file: VGPRs, spills, MFMA count, and every `s_waitcnt vmcnt(0)` that sits BETWEEN two MFMAs (a drain of the
in-order load / store queue in the middle of a tile).   usage: python tools/scan_kernel_waits.py file.s"""
import re
import sys

for path in sys.argv[1:]:
    name, m, hits, total = None, 0, [], {}
    for ln, line in enumerate(open(path), 1):
        mm = re.match(r"^(_Z\w+):", line)
        if mm:
            if name:
                total[name] = (m, hits)
            name, m, hits = mm.group(1), 0, []
        if "v_mfma" in line:
            m += 1
        elif "s_waitcnt" in line and "vmcnt(0)" in line and m > 0:
            hits.append((ln, m))
        elif name and ".end_amdhsa_kernel" in line:
            total[name] = (m, hits)
            name = None
    meta = dict(re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", open(path).read()))
    for k, (m, hits) in total.items():
        inside = [h for h in hits if h[1] < m]
        print(f"{k[:70]:70s} mfma {m:4d}  vgpr {meta.get(k, '?'):>3s}  vmcnt(0) between MFMAs: {len(inside)} {[h[1] for h in inside][:12]}")
