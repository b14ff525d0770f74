"""Average of every collected rocprofv3 --pmc counter per kernel name: counter_collection.csv (one row per dispatch and counter) -> a
CSV with one row per kernel. Usage: pmc_kernel_summary.py counter_collection.csv [name filter] > summary.csv"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if flt and flt not in n:
            continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for d in acc.values() for c in d})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches"] + ["avg_" + c for c in counters])
    for n, d in sorted(acc.items()):
        w.writerow([n[:120], max(len(v) for v in d.values())] + ["%.1f" % (sum(d[c]) / len(d[c])) if d.get(c) else "" for c in counters])


if __name__ == "__main__":
    main()
