#!/bin/bash
# Round profiles: rocprofv3 kernel statistics of the bench command per model (in-situ durations, two graph lanes), HBM traffic
# per launch from separate --pmc passes (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE), and an SQ pass with the
# matrix-pipe busy counter.  usage: tools/profile_session.sh TAG [ROUND]  ->  gpurun_out/TAG/{MODEL}_*  (copy what matters to profiles/ROUND)
TAG=${1:-prof}; export ROUND=${2:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for spec in resnet50:256 vit_base:256 swin_t:128 alexnet:256; do
  M=${spec%%:*}; B=${spec##*:}
  CMD="python bench.py --model $M --batch $B --steps 20 --warmup 5 --no-cpu --extra none --soak 1 --no-lanes1"
  timeout 400 $CMD --layers $O/${M}_per_launch.txt > $O/${M}_bench_layers.json 2> $O/${M}_bench.err
  # THE profiled command (its own JSON line is what roofline.frac is checked against): no --layers, no lanes1 pass -> the trace
  # holds only the two-lane launches (graph replays + the in-situ event pass)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${M}_trace -o t -- $CMD > $O/${M}_trace.log 2>&1
  grep '^{' $O/${M}_trace.log | tail -1 > $O/${M}_bench.json
  t=$(find $O/${M}_trace -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/rocprof_trim.py $t $O/${M}_rocprofv3_warm_stats.txt
  s=$(find $O/${M}_trace -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $O/${M}_rocprofv3_kernel_stats.csv
  # the same model as ONE launch list (--lanes 1): per-kernel durations whose sum compares with a step
  CMD1="python bench.py --model $M --batch $B --steps 20 --warmup 5 --no-cpu --extra none --soak 1 --lanes 1"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${M}_trace1 -o t -- $CMD1 > $O/${M}_trace1.log 2>&1
  grep '^{' $O/${M}_trace1.log | tail -1 > $O/${M}_lanes1_bench.json
  t=$(find $O/${M}_trace1 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/rocprof_trim.py $t $O/${M}_lanes1_rocprofv3_warm_stats.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${M}_$C -o t -- python bench.py --model $M --batch $B --steps 3 --warmup 2 --no-cpu --extra none --soak 0 --no-lanes1 > $O/${M}_$C.log 2>&1
  done
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${M}_SQ -o t -- python bench.py --model $M --batch $B --steps 2 --warmup 2 --no-cpu --extra none --soak 0 --no-graph --no-lanes1 > $O/${M}_SQ.log 2>&1
done
for spec in resnet50 vit_base swin_t alexnet; do
  t=$(find $O/${spec}_trace -name "*kernel_trace.csv" | head -1)
  n=$(python -c "import json; print(json.load(open('$O/${spec}_bench.json'))['config']['launches_per_step'])")
  [ -n "$t" ] && python tools/graph_timeline.py $t $n 7 > $O/${spec}_graph_timeline_under_rocprofv3.txt
done
find $O -name "*.db" -delete
O=$O python - <<'PY'
import csv, glob, collections, json, os, re, statistics
O = os.environ["O"]
import sys; sys.path.insert(0, 'tools')
from kernel_families import fam
out = {"_batch": {}, "_rocprof": {}, "_rocprof_lanes1": {}}
lines = []
for M, B in (("resnet50", 256), ("vit_base", 256), ("swin_t", 128), ("alexnet", 256)):
    acc = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
    for C in acc:
        for f in glob.glob(f"{O}/{M}_{C}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == C:
                    k = fam(r["Kernel_Name"])
                    if k: acc[C][k].append(float(r["Counter_Value"]))
    tr = {}
    for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
        fs, ws = acc["FETCH_SIZE"].get(k, []), acc["WRITE_SIZE"].get(k, [])
        rd = 2.0 * 1024.0 * sum(fs) / max(1, len(fs))           # KB, doubled (guide: gfx950 FETCH_SIZE = half of a coalesced stream)
        wr = 1024.0 * sum(ws) / max(1, len(ws))
        tr[k] = round(rd + wr)
        lines.append(f"{M:9s} {k:34s} launches {len(fs):5d}  read {rd/1e6:9.2f} MB  write {wr/1e6:9.2f} MB  total {(rd+wr)/1e6:9.2f} MB per launch")
    if tr:
        out[M] = tr
        out["_batch"][M] = B
    # in-situ kernel durations (two graph lanes overlap): warm average per family
    rp = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{M}_trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = fam(r["Kernel_Name"])
            if k: rp[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out["_rocprof"][M] = {}
    for k, v in rp.items():
        w = sorted(v)[: max(1, len(v) - max(1, len(v) // 50))]
        out["_rocprof"][M][k] = {"avg_launch_us": round(sum(w) / len(w), 2), "calls": len(v), "file": f"profiles/{os.environ.get('ROUND', 'r03')}/{M}_rocprofv3_warm_stats.txt"}
    # the same for the ONE-lane command (full-batch launches back to back): what bench.py's headline roofline is compared with
    rp1 = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{M}_trace1/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = fam(r["Kernel_Name"])
            if k: rp1[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out["_rocprof_lanes1"][M] = {}
    for k, v in rp1.items():
        w = sorted(v)[: max(1, len(v) - max(1, len(v) // 50))]
        out["_rocprof_lanes1"][M][k] = {"avg_launch_us": round(sum(w) / len(w), 2), "calls": len(v),
                                        "file": f"profiles/{os.environ.get('ROUND', 'r04')}/{M}_lanes1_rocprofv3_warm_stats.txt"}
    # SQ pass: MFMA busy fraction per family (busy cycles per SIMD / kernel cycles)
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{M}_SQ/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = fam(r["Kernel_Name"])
            if k: sq[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(f"{O}/{M}_pmc_sq.txt", "w") as g:
        g.write(f"# {M}: SQ counters per launch (mean over launches); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 SEs)\n")
        for k, d in sorted(sq.items()):
            m = {c: sum(v) / len(v) for c, v in d.items()}
            dur = m.get("SQ_BUSY_CYCLES", 0) / 32.0
            frac = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / dur if dur else 0
            g.write(f"{k:34s} n={len(next(iter(d.values()))):4d} mfma_busy_frac {frac:5.3f}  kernel_cycles {dur:12.0f}  " +
                    "  ".join(f"{c}={m[c]:.3g}" for c in sorted(m)) + "\n")
# bench.py's live roofline figure against the trace of the same command.  The trace holds (a) the graph replays and (b) at its end
# bench.py's in-situ pass: 7 eager two-stream replays of the launch lists (the last 6 are averaged).  (b) is the like-for-like
# comparison: same launches, same concurrency, two clocks.  (a) differs: under rocprofv3 hipGraphLaunch enqueues branch after branch
# so slowly that the second lane starts when the first is half done (<model>_graph_timeline_under_rocprofv3.txt).
agree = ["# roofline.avg_launch_us printed by bench.py (HIP events, live) vs the rocprofv3 kernel trace of the SAME command:",
         "# 'same pass' = the trace rows of bench.py's own in-situ pass (last 6 eager replays); 'graph replays' = warm average over the whole trace"]
for M in ("resnet50", "vit_base", "swin_t", "alexnet"):
    try:
        bj = json.load(open(f"{O}/{M}_bench.json"))
        r = bj["roofline"]
        rp = out["_rocprof"][M][r["kernel"]]
        N = bj["config"]["launches_per_step"]
        rows = []
        for f in glob.glob(f"{O}/{M}_trace/**/*kernel_trace.csv", recursive=True):
            for x in csv.DictReader(open(f)):
                rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"]))
        rows.sort()
        names = [k for _, _, k in rows]
        for P in range(N, 4 * N):                       # kernels per replay: a C-ABI call may launch more than one kernel
            if names[-P:] == names[-2 * P:-P] and sorted(names[-P:]) == sorted(names[-3 * P:-2 * P]):
                N = P
                break
        win = [(e - s) / 1e3 for s, e, k in rows[-6 * N:] if fam(k) == r["kernel"]]
        same = sum(win) / max(1, len(win))
        ok = len(win) == 6 * r["launches_per_step"]
        agree.append(f"{M:9s} {r['kernel']:30s} bench {r['avg_launch_us']:8.2f} us | same pass {same:8.2f} us ({len(win)} rows{'' if ok else ' (!) expected ' + str(6 * r['launches_per_step'])}) "
                     f"ratio {r['avg_launch_us'] / same if same else 0:.3f} | graph replays {rp['avg_launch_us']:8.2f} us ({rp['calls']} calls)   frac {r['frac']}")
    except Exception as e:
        agree.append(f"{M}: {type(e).__name__} {e}")
agree.append("# headline roofline (one lane): bench.py --lanes 1 under rocprofv3, its own avg_launch_us (HIP events) vs the trace's warm average of the same kernel")
for M in ("resnet50", "vit_base", "swin_t", "alexnet"):
    try:
        bj = json.load(open(f"{O}/{M}_lanes1_bench.json"))
        r = bj["roofline"]
        rp = out["_rocprof_lanes1"][M][r["kernel"]]
        fl = r["flop_per_launch"]
        agree.append(f"{M:9s} {r['kernel']:34s} events {r['avg_launch_us']:8.2f} us (frac {r['frac']}) | rocprofv3 {rp['avg_launch_us']:8.2f} us "
                     f"({rp['calls']} calls, frac {fl / rp['avg_launch_us'] / 1e6 / 2500:.4f}) ratio {r['avg_launch_us'] / rp['avg_launch_us']:.3f}")
    except Exception as e:
        agree.append(f"{M} lanes1: {type(e).__name__} {e}")
open(f"{O}/roofline_vs_rocprof.txt", "w").write("\n".join(agree) + "\n")
print("\n".join(agree))
json.dump(out, open(f"{O}/traffic.json", "w"), indent=1)
open(f"{O}/hbm_traffic_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
os.system(f"find {O} -size +3M -delete")
PY
cat $O/*_bench.json | cut -c1-400
cat $O/*_pmc_sq.txt | cut -c1-200
