"""Markdown tables for DESIGN.md section 5 out of the committed profiles: per model, the kernel families by share of kernel time with
duration alone / in situ, achieved TFLOP/s and GB/s (alone), algorithmic vs PMC-counted HBM bytes and the matrix-pipe-busy fraction.
usage: design_tables.py [ROUND=r03]"""
import json, os, re, sys, collections
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
tj = json.load(open(os.path.join(P, "traffic.json")))
_out = []
def print(*a):                                                  # noqa: A001 -- collect, then patch DESIGN.md between its markers
    _out.append(" ".join(str(x) for x in a))
for M in ("resnet50", "vit_base", "swin_t", "alexnet"):
    b = json.load(open(f"{P}/{R}/{M}_bench_layers.json"))
    bp = json.load(open(f"{P}/{R}/{M}_bench.json"))
    b1 = json.load(open(f"{P}/{R}/{M}_lanes1_bench.json"))
    r = b["roofline"]; wf = r["whole_forward"]
    fam = collections.OrderedDict()
    for line in open(f"{P}/{R}/{M}_per_launch.txt"):
        if line.startswith("#") or line.startswith("kernel"):
            continue
        k = line.split()[0]                                     # one row per kernel name = one kernel symbol
        us, tf, gbs, gf, mb, pct = [float(x) for x in line.split()[-6:]]
        d = fam.setdefault(k, dict(n=0, us=0.0, gf=0.0, mb=0.0))
        d["n"] += 1; d["us"] += us; d["gf"] += gf; d["mb"] += mb
    sq = {}
    for line in open(f"{P}/{R}/{M}_pmc_sq.txt"):
        m = re.match(r"(\S+)\s+n=\s*\d+ mfma_busy_frac\s+([\d.]+)", line)
        if m: sq[m.group(1)] = float(m.group(2))
    rp = tj["_rocprof"].get(M, {})
    tot = sum(d["us"] for d in fam.values())
    print(f"\n**{M}** (B = {tj['_batch'][M]}, two lanes): **{b['value'] / 1e3:.1f} k img/s**, {b['ms_per_step']:.2f} ms/step, {b['config']['launches_per_step']} launches/step; "
          f"whole forward {wf['frac']:.3f} of the MFMA peak, {wf['hbm_frac']:.3f} of the HBM peak on {wf['algorithmic_mb'] / 1e3:.2f} GB algorithmic; "
          f"one lane: {b1['value'] / 1e3:.1f} k; under rocprofv3: {bp['value'] / 1e3:.1f} k.\n")
    print("| kernel family | launches / step | share of kernel time | us alone | us in situ (rocprofv3) | TFLOP/s alone (frac of 2500) | GB/s alone (frac of 8000) | algorithmic MB | PMC MB (ratio) | matrix pipe busy |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for k, d in sorted(fam.items(), key=lambda kv: -kv[1]["us"])[:8]:
        a = d["us"] / d["n"]
        ins = rp.get(k, {}).get("avg_launch_us")
        pm = tj.get(M, {}).get(k)
        amb = d["mb"] / d["n"]
        tf = d["gf"] / d["us"] * 1e3 if d["us"] else 0
        gb = d["mb"] / d["us"] * 1e3 if d["us"] else 0
        print(f"| `{k}` | {d['n']} | {100 * d['us'] / tot:.0f} % | {a:.1f} | {ins if ins else '-'} | {tf:.0f} ({tf / 2500:.2f}) | {gb:.0f} ({gb / 8000:.2f}) | {amb:.0f} | "
              + (f"{pm / 1e6:.0f} ({pm / 1e6 / amb:.2f}x)" if pm and amb else "-") + f" | {sq.get(k, '-')} |")

import builtins
text = "\n".join(_out).strip("\n")
builtins.print(text)
D = os.path.join(os.path.dirname(P), "DESIGN.md")
d = open(D).read()
b, e = "<!-- tables:begin -->", "<!-- tables:end -->"
if b in d and e in d:
    d = d[:d.index(b) + len(b)] + "\n" + text + "\n" + d[d.index(e):]
    open(D, "w").write(d)
