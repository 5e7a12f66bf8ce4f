"""From a rocprofv3 kernel trace of `bench.py`: fraction of the timed graph replays' span with at least one kernel running, and the
average number of kernels in flight (two graph lanes).  usage: trace_gaps.py <kernel_trace.csv> <launches per step> <timed steps>
(the per-launch worksheet replay at the end of a bench run -- 6 launches per recorded call -- is cut off)"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
n = len(rows)
lps, steps = int(sys.argv[2]), int(sys.argv[3])
tail = 6 * lps
rows = rows[n - tail - lps * steps: n - tail]
span = rows[-1][1] - rows[0][0]
busy, (cs, ce) = 0, rows[0][:2]
gaps = []
for s, e, _ in rows[1:]:
    if s <= ce:
        ce = max(ce, e)
    else:
        busy += ce - cs
        gaps.append(s - ce)
        cs, ce = s, e
busy += ce - cs
tot = sum(e - s for s, e, _ in rows)
gaps.sort()
print(f"launches {len(rows)}, span {span / 1e6:.2f} ms: a kernel is running {100.0 * busy / span:.1f}% of the time; "
      f"sum of durations / span = {tot / span:.2f}; idle gaps: {len(gaps)} (median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:.2f} us, "
      f"total {sum(gaps) / 1e6:.3f} ms)")
