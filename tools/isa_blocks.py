#!/usr/bin/env python3
"""tools/isa_blocks.py <asm.s> <kernel-symbol-prefix> [min] — static instruction mix per basic block of one kernel
(`hipcc -S --cuda-device-only` assembly): VALU / SALU / LDS / vector-memory counts, loop depth as the assembler annotates it.
Blocks with fewer than `min` (default 12) instructions are folded into the totals only."""
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    least = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith(sym) and l.split(";")[0].rstrip().endswith(":")][0]
    end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
    blocks, cur = [], None
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            cur = [m.group(1), (m.group(2) or "").strip("; ").strip(), 0, 0, 0, 0, 0]
            blocks.append(cur)
            continue
        t = l.strip()
        if not t or t[0] in ";.":
            continue
        if cur is None:
            cur = ["entry", "", 0, 0, 0, 0, 0]
            blocks.append(cur)
        op = t.split()[0]
        if op.startswith("v_"):
            cur[2] += 1
        elif op.startswith("s_"):
            cur[3] += 1
        elif op.startswith("ds_"):
            cur[4] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            cur[5] += 1
        cur[6] += 1
    tot = [sum(b[i] for b in blocks) for i in (2, 3, 4, 5, 6)]
    print("static: VALU %d  SALU %d  LDS %d  VMEM %d  total %d  blocks %d" % (*tot, len(blocks)))
    for b in blocks:
        if b[6] >= least:
            print("%-12s V%-4d S%-4d L%-3d M%-3d %s" % (b[0], b[2], b[3], b[4], b[5], b[1][:70]))


if __name__ == "__main__":
    main()
