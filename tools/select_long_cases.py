#!/usr/bin/env python3
"""Which CPU test cases the default `pytest -m "not gpu"` run leaves out (tests/long_cases.txt).

The CPU suite executes every HIP kernel of the library on the emulated device (tests/emu); all of it is ~100 CPU-minutes, and the
default run has to finish in a few minutes. Input: the `--durations=0` output of a full run
    python -m pytest tests -q -m "not gpu" --all-cases -n 8 --durations=0 > durations.log
Selection: cases are kept in order of increasing duration until they add up to BUDGET seconds; on top of that every test FUNCTION
that would disappear keeps its cheapest case when that case takes at most KEEP_ONE seconds. Everything else goes to tests/long_cases.txt, which
tests/conftest.py reads: those cases are skipped unless --all-cases (or CSEG_TESTS_ALL=1) is given.
    python tools/select_long_cases.py durations.log [budget_seconds] > tests/long_cases.txt
"""
import re
import sys
from collections import defaultdict

KEEP_ONE = 15.0
# never skipped, whatever they cost: the oracle against the reference's golden vectors (what pins the oracle), the C-ABI surface, and
# one whole 2-rank DDP trainer step over gloo
ALWAYS = (r"tests/test_oracle_", r"tests/test_cabi\.py", r"::test_trainer_ddp_two_ranks_keeps_replicas_in_sync")


def main():
    log = sys.argv[1]
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    dur = defaultdict(float)
    for line in open(log):
        m = re.match(r"\s*([0-9.]+)s (call|setup|teardown)\s+(.+?)\s*$", line)          # (node ids may contain blanks)
        if m:
            dur[m.group(3)] += float(m.group(1))
    by_func = defaultdict(list)
    for nid, d in dur.items():
        by_func[nid.split("[")[0]].append((d, nid))
    keep = {nid for nid in dur if any(re.search(p, nid) for p in ALWAYS)}
    total = sum(dur[nid] for nid in keep)
    for d, nid in sorted((d, nid) for nid, d in dur.items() if nid not in keep):          # cheap cases first, up to the budget
        if total + d > budget:
            break
        keep.add(nid)
        total += d
    for func, cases in by_func.items():                                 # on top: no function disappears if one case is affordable
        d, nid = min(cases)
        if d <= KEEP_ONE and not any(n in keep for _, n in cases):
            keep.add(nid)
            total += d
    long_cases = sorted(nid for nid in dur if nid not in keep)
    print("# cases the default CPU run skips (tools/select_long_cases.py; run them with --all-cases or CSEG_TESTS_ALL=1)")
    print("# kept: %d cases, %.0f s of test time; skipped: %d cases, %.0f s" %
          (len(keep), total, len(long_cases), sum(dur[n] for n in long_cases)))
    for nid in long_cases:
        print("%s\t%.1f" % (nid, dur[nid]))


if __name__ == "__main__":
    main()
