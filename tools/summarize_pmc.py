"""Averages a rocprofv3 --pmc counter_collection.csv per kernel name.
Usage: summarize_pmc.py file.csv [COUNTER]   (no counter: every counter in the file, one column each)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else None
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    c = r.get("Counter_Name")
    if want is None or c == want:
        acc[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
print("kernel,dispatches," + ",".join("avg_%s" % c for c in counters))
for k, v in sorted(acc.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values())):
    n = max(len(x) for x in v.values())
    print('"%s",%d,%s' % (k[:120], n, ",".join("%.1f" % (sum(v[c]) / len(v[c])) if v.get(c) else "" for c in counters)))
