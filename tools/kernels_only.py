"""Runs only the hand-written HIP kernels at the bench workload's shapes (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    out = bench.kernel_rooflines(torch.device("cuda", 0), 8)
    for k, v in out.items():
        print(k, v)
