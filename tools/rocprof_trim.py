"""Per-kernel launch statistics from a rocprofv3 kernel trace, with the cold first launches set aside.
rocprofv3's own *_kernel_stats.csv averages in the first launch of every kernel (module load, 10-25 ms), which
skews AverageNs for kernels launched a few hundred times; this prints both.   usage: rocprof_trim.py TRACE.csv [OUT.txt]"""
import csv, sys, collections, statistics

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
tot = sum(sum(v) for v in rows.values())
out.write(f"{'kernel':86s} {'calls':>6s} {'avg_us':>9s} {'avg_us_warm':>11s} {'median_us':>9s} {'min_us':>8s} {'max_us':>10s} {'%time_warm':>10s}\n")
warm_tot = 0.0
stats = []
for k, v in rows.items():
    w = sorted(v)[: max(1, len(v) - max(1, len(v) // 50))]        # drop the slowest 2% (at least one) launches
    stats.append((k, v, w))
    warm_tot += sum(w)
for k, v, w in sorted(stats, key=lambda t: -sum(t[2])):
    out.write(f"{k[:86]:86s} {len(v):6d} {sum(v)/len(v)/1e3:9.2f} {sum(w)/len(w)/1e3:11.2f} {statistics.median(v)/1e3:9.2f} "
              f"{min(v)/1e3:8.2f} {max(v)/1e3:10.2f} {100*sum(w)/warm_tot:10.2f}\n")
