"""Static instruction mix of a compiled gfx950 kernel, per basic block: python tools/asm_blocks.py file.s <substring of the kernel symbol>
(the .s comes from `hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only`). Columns: MFMA, other VALU, LDS reads / writes, global +
buffer loads / stores, SALU, waitcnt, barriers. Blocks with no MFMA and fewer than 8 instructions are folded into 'other'."""
import re
import sys
from collections import OrderedDict


def classify(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_r"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_w"
    if op.startswith("ds_"):
        return "lds_o"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vm_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
        return "vm_st"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "misc"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = {}
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = {}
            continue
        s = l.strip()
        if not s or s.startswith((";", ".")):
            continue
        op = s.split()[0]
        c = classify(op)
        blocks[cur][c] = blocks[cur].get(c, 0) + 1
        if c == "wait":
            blocks[cur].setdefault("waits", []).append(s.split(None, 1)[1].split(";")[0].strip())
        if op.startswith(("s_cbranch", "s_branch")):
            blocks[cur].setdefault("br", []).append(s.split()[-1])
    cols = ["mfma", "valu", "lds_r", "lds_w", "vm_ld", "vm_st", "salu", "wait", "barrier"]
    print("%-14s" % "block" + "".join("%7s" % c for c in cols) + "  branches / waits")
    tot = {}
    for name, b in blocks.items():
        n = sum(v for k, v in b.items() if isinstance(v, int))
        for c in cols:
            tot[c] = tot.get(c, 0) + b.get(c, 0)
        if n < 8 and not b.get("mfma"):
            continue
        print("%-14s" % name + "".join("%7d" % b.get(c, 0) for c in cols) + "  " + ",".join(b.get("br", [])) + "  " +
              " | ".join(b.get("waits", []))[:150])
    print("%-14s" % "total" + "".join("%7d" % tot.get(c, 0) for c in cols))


if __name__ == "__main__":
    main()
