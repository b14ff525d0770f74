"""Per-kernel statistics of the LAST `window_s` seconds of a rocprofv3 kernel trace (i.e. the timed steady-state
steps of bench.py, excluding MIOpen's first-use find and the warm-up). Usage: trace_window_stats.py trace.csv window_s"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e9
t_end = max(int(r["End_Timestamp"]) for r in rows)
acc = defaultdict(lambda: [0, 0])
tot = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t_end - win:
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += e - s
        tot += e - s
print("Name,Calls,TotalDurationNs,AverageNs,Percentage")
for k, (n, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%d,%.1f,%.2f' % (k.replace('"', "'"), n, d, d / n, 100.0 * d / tot))
sys.stderr.write("window %.3f s: %d dispatches, %.1f ms of kernel time\n" % (win / 1e9, sum(v[0] for v in acc.values()), tot / 1e6))
