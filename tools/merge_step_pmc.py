"""profiles/r03_step_pmc.json from the two per-counter summaries of tools/step_pmc_summary.py (FETCH_SIZE and WRITE_SIZE are collected
in separate rocprofv3 passes, MI355X_MICROARCH.md). Usage: merge_step_pmc.py FETCH.json WRITE.json "<what was measured>" > out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


from merge_step_pmc_stamp import tree_stamp

f, w = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
fb, wb = f["per_step_KB"] * 1024, w["per_step_KB"] * 1024
print(json.dumps({
    "tree_stamp": tree_stamp(),
    "hbm_bytes_per_step": int(fb + wb), "fetch_bytes_per_step": int(fb), "write_bytes_per_step": int(wb),
    "fetch_bytes_per_step_x2": int(2 * fb), "by_family_fetch_KB": f["per_step_KB_by_family"],
    "by_family_write_KB": w["per_step_KB_by_family"], "dispatches_per_step": f["dispatches_per_step"],
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --steps 2 --warmup 2 --no-cpu-baseline "
            "--no-kernels --no-fp32-pass`, summed over the kernels of one steady-state step (tools/step_pmc_summary.py), N=1, "
            + sys.argv[3] + "; hbm_bytes = raw FETCH_SIZE + WRITE_SIZE. Per MI355X_MICROARCH.md gfx950 FETCH_SIZE counts 64 B per "
            "128-B request for wide coalesced reads (16 B per lane), which is what the split kernels' LDS-DMA weight stream, the BN "
            "kernels and the elementwise kernels issue: the true read traffic lies between fetch and 2 x fetch "
            "(fetch_bytes_per_step_x2; MIOpen's kernels uncalibrated). Algorithmic activation traffic of the step: ~192 GB (24 GB "
            "per image, BASELINE.md section 2)"}, indent=1))
