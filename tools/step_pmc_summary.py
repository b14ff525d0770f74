"""Per-step totals of one rocprofv3 --pmc counter (FETCH_SIZE / WRITE_SIZE, KB) over the steady-state steps of a
bench.py run. Steps are delimited by a kernel that runs exactly once per step (default: ce_finish_kernel of the fused
CE forward); the last `steps` complete marker-to-marker spans are averaged. Also prints the split by kernel family.
Usage: step_pmc_summary.py counter_collection.csv COUNTER [steps] [marker]  ->  JSON on stdout"""
import csv
import json
import sys
from collections import defaultdict


def family(n):
    if "transpose" in n:
        return "miopen_layout_transpose"
    if "BatchNorm" in n or "bn_" in n:
        return "batchnorm"
    if n.startswith("igemm") or "Conv" in n or "conv" in n or "gemm" in n.lower() or n.startswith("Cijk") or "Col" in n:
        return "conv_gemm"
    if "at::native" in n or "SubTensorOp" in n or "rocclr" in n:
        return "torch_elementwise_copy"
    return "cseg_hip_kernels"


def main():
    path, want = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    marker = sys.argv[4] if len(sys.argv) > 4 else "ce_finish_kernel"
    rows = []
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == want:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    marks = [i for i, (_, n, _) in enumerate(rows) if marker in n]
    if len(marks) < steps + 1:
        raise SystemExit("only %d markers for %d steps" % (len(marks), steps))
    lo, hi = marks[-(steps + 1)], marks[-1]
    fam = defaultdict(float)
    for _, n, v in rows[lo:hi]:
        fam[family(n)] += v
    total = sum(fam.values())
    print(json.dumps({"counter": want, "unit": "KB", "steps_averaged": steps, "dispatches_per_step": (hi - lo) / steps,
                      "per_step_KB": total / steps,
                      "per_step_KB_by_family": {k: round(v / steps, 1) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}}))


if __name__ == "__main__":
    main()
