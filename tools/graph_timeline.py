"""Timeline of ONE hipGraph replay out of a rocprofv3 kernel trace: start / end / duration of every kernel of the last complete
replay, so that the overlap pattern of the two lanes can be read off.  usage: graph_timeline.py kernel_trace.csv LAUNCHES_PER_STEP [TRAILING_STEPS_TO_SKIP]
(bench.py replays the launch lists eagerly for 7 steps after the timed graph replays: skip 7 to land on the last graph replay)"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
n = int(sys.argv[2])
skip = int(sys.argv[3]) * n if len(sys.argv) > 3 else 0
last = rows[len(rows) - skip - n:len(rows) - skip]
t0 = last[0][0]
print(f"# one hipGraph replay ({n} kernels) out of {sys.argv[1].split('/')[-1]}: start us, end us, duration, queue, stream, kernels overlapping (ovN), name")
print(f"# span {(max(e for _, e, *_ in last) - t0) / 1e3:.1f} us; some kernel of the OTHER queue in flight during {sum(1 for s, e, k, q, st in last if any(q2 != q and s2 < e and e2 > s for s2, e2, k2, q2, st2 in last))} of {n} kernels")
for s, e, k, q, st in last:
    ov = sum(1 for s2, e2, *_ in last if s2 < e and e2 > s) - 1
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} us  q{q:>3s} s{st:>3s} ov{ov}  {k}")
