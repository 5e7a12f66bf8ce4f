"""profiles/traffic.json must follow from the files it cites (round-4 review, weak 6b): every `_rocprof` / `_rocprof_lanes1`
entry names a committed `*_warm_stats.txt` (tools/rocprof_trim.py of a rocprofv3 kernel trace) and an `avg_launch_us`; the same
figure is re-derived here from that file's rows through the one symbol -> family table (tools/kernel_families.py).  A family made
of ONE kernel symbol must match its row's avg_us_warm to the cent; a family that spans several template instances (stream1x1 ...)
was averaged over the pooled launches, so its figure has to lie between its rows' extremes."""
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_families import fam  # noqa: E402

ROW = re.compile(r"^(?P<name>.{86}) +(?P<calls>\d+) +(?P<avg>[\d.]+) +(?P<warm>[\d.]+) +(?P<med>[\d.]+) +(?P<min>[\d.]+) +(?P<max>[\d.]+) +(?P<pct>[\d.]+)$")


def _rows(path):
    fams = {}
    for line in open(path).read().splitlines()[1:]:
        m = ROW.match(line)
        if not m:
            continue
        k = fam(m.group("name"))
        if k:
            fams.setdefault(k, []).append((int(m.group("calls")), float(m.group("warm"))))
    return fams


def _entries():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for table in ("_rocprof", "_rocprof_lanes1"):
        for model, d in tj.get(table, {}).items():
            for k, e in d.items():
                yield table, model, k, e


def test_traffic_json_cites_existing_files():
    seen = set()
    for table, model, k, e in _entries():
        assert os.path.exists(os.path.join(ROOT, e["file"])), f"{table}.{model}.{k} cites {e['file']}, which is not committed"
        seen.add(e["file"])
    assert seen, "profiles/traffic.json has no _rocprof tables"


def test_rocprof_tables_follow_from_the_cited_stats():
    cache = {}
    checked = 0
    for table, model, k, e in _entries():
        rows = cache.setdefault(e["file"], _rows(os.path.join(ROOT, e["file"])))
        assert k in rows, f"{table}.{model}.{k}: no row of {e['file']} maps to this family"
        r = rows[k]
        assert sum(c for c, _ in r) == e["calls"], f"{table}.{model}.{k}: calls {e['calls']} vs {r} in {e['file']}"
        if len(r) == 1:
            assert abs(r[0][1] - e["avg_launch_us"]) <= 0.011, f"{table}.{model}.{k}: {e['avg_launch_us']} us vs avg_us_warm {r[0][1]} in {e['file']}"
        else:
            lo, hi = min(w for _, w in r), max(w for _, w in r)
            assert lo * 0.97 <= e["avg_launch_us"] <= hi * 1.03, f"{table}.{model}.{k}: {e['avg_launch_us']} us outside {lo}..{hi} ({e['file']})"
        checked += 1
    assert checked > 20


def test_traffic_batches_match_the_bench_defaults():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert tj["_batch"] == {"resnet50": 256, "vit_base": 256, "swin_t": 128, "alexnet": 256}
    for model in tj["_batch"]:
        assert tj.get(model), f"no PMC traffic table for {model}"
