"""Averages a rocprofv3 --pmc counter_collection.csv per kernel name. Usage: summarize_pmc.py file.csv COUNTER"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2]
acc = defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == want:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print("kernel,dispatches,avg_%s" % want)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%.1f' % (k[:120], len(v), sum(v) / len(v)))
