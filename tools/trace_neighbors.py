"""Kernels dispatched right before / after the long launches of a given kernel (substring match) in the last `window_s` seconds
of a rocprofv3 kernel trace: tells which host-side operation an anonymous elementwise / copy kernel belongs to.
Usage: trace_neighbors.py trace.csv window_s substring [min_us] [context]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e9
pat = sys.argv[3]
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 100.0
ctx = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
t_end = max(e for _, e, _ in ev)
ev = [x for x in ev if x[0] >= t_end - win]
shown = 0
for i, (s, e, n) in enumerate(ev):
    if pat in n and (e - s) / 1e3 >= min_us:
        print("---- %s  %.1f us at %.2f ms" % (n[:90], (e - s) / 1e3, (s - ev[0][0]) / 1e6))
        for j in range(max(0, i - ctx), min(len(ev), i + ctx + 1)):
            print("   %s %8.1f us  %s" % (">>" if j == i else "  ", (ev[j][1] - ev[j][0]) / 1e3, ev[j][2][:110]))
        shown += 1
        if shown >= 6:
            break
