#!/usr/bin/env python3
"""tools/isa_scratch.py <asm.s> <kernel-symbol> — every scratch (private memory) access of a kernel with the basic block it sits in and
whether that block belongs to a loop (the assembler's "in Loop" / "Loop Header" annotations). The packet loops of the fused synthesis
kernels must hold none: a spill or reload is a vector-memory operation, queued in order behind the residue look-ahead."""
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
on, label, loop = False, "entry", ""
inloop = outloop = 0
for l in open(path):
    if l.startswith(sym + ":"):
        on = True
        continue
    if not on:
        continue
    if "s_endpgm" in l or l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if m:
        label = m.group(1)
        c = m.group(2) or ""
        loop = "loop" if ("in Loop" in c or "Loop Header" in c or "Inner Loop" in c) else ""
        continue
    if re.match(r"^; %bb\.\d+:\s*;.*(in Loop|Loop Header)", l):
        loop = "loop"
    elif re.match(r"^; %bb\.\d+:", l):
        loop = ""
    if "scratch_" in l:
        print("%-5s %-12s %s" % (loop, label, " ".join(l.split()[:6])))
        if loop:
            inloop += 1
        else:
            outloop += 1
print("scratch accesses: %d inside loops, %d outside" % (inloop, outloop))
