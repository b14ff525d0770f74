"""tree_stamp(): sha256 (16 hex digits) over the kernel sources and kernels.py -- what decides the launches of a step. Written into
profiles/rNN_step_pmc.json by tools/merge_step_pmc.py and compared by bench.py (a counter file of another tree is refused)."""
import glob
import hashlib
import os


def tree_stamp():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "contrastiveseg_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "contrastiveseg_amd", "csrc", "*.h"))
                   + [os.path.join(root, "contrastiveseg_amd", "kernels.py")])
    h = hashlib.sha256()
    for f_ in files:
        h.update(open(f_, "rb").read())
    return h.hexdigest()[:16]
