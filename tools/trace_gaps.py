"""Largest idle gaps between consecutive kernel dispatches in the last `window_s` seconds of a rocprofv3 kernel trace.
Usage: trace_gaps.py trace.csv window_s [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e9
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
t_end = max(e for _, e, _ in ev)
ev = [x for x in ev if x[0] >= t_end - win]
gaps = []
busy_end = ev[0][1]
prev = ev[0][2]
idle = 0
for s, e, n in ev[1:]:
    if s > busy_end:
        gaps.append((s - busy_end, prev, n, (s - ev[0][0]) / 1e6))
        idle += s - busy_end
    if e > busy_end:
        busy_end = e
        prev = n
span = ev[-1][1] - ev[0][0]
print("window span %.1f ms, idle %.1f ms (%.1f%%), %d gaps > 20us" % (span / 1e6, idle / 1e6, 100.0 * idle / span,
                                                                      sum(1 for g in gaps if g[0] > 20000)))
for g in sorted(gaps, reverse=True)[:top]:
    print("%8.1f us  at %8.1f ms   after %-60s before %-60s" % (g[0] / 1e3, g[3], g[1][:60], g[2][:60]))
