#!/usr/bin/env python3
"""tools/isa_waits.py <asm.s> <kernel-symbol> — where does a kernel wait for vector memory?
Input: `hipcc -S --cuda-device-only` assembly. Prints, in program order, every vector-memory instruction group and every
s_waitcnt with a vmcnt component, with the basic-block label and whether the block belongs to a depth-1 loop. The packet loop of
the fused synthesis kernel must only hold COUNTED waits (vmcnt(N), N > 0) outside its rare blocks: a vmcnt(0) there waits for the
residue prefetch and the previous packet's PCM stores (see vmem_drain() in csrc/vsyn_fused.h)."""
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    on, label, loop = False, "", ""
    prev_kind, run = None, 0

    def flush():
        nonlocal prev_kind, run
        if prev_kind:
            print("%-14s %-44s x%d" % (loop, label, run), prev_kind)
        prev_kind, run = None, 0

    for ln, l in enumerate(open(path), 1):
        if l.startswith(sym + ":"):
            on = True
            continue
        if not on:
            continue
        if "s_endpgm" in l:
            break
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            flush()
            label = m.group(1)
            c = m.group(2) or ""
            h = re.search(r"Header=(BB\d+_\d+)", c)
            loop = "loop " + h.group(1) if h else ("LOOP HEAD" if "Loop Header" in c else "")
            continue
        t = l.strip()
        kind = None
        if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", t):
            kind = t.split()[0]
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            kind = t
        if kind != prev_kind:
            flush()
            prev_kind, run = kind, 0
        if kind:
            run += 1
    flush()


if __name__ == "__main__":
    main()
