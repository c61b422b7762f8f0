#!/usr/bin/env python3
"""tools/isa_loops.py — per-loop instruction histogram of one kernel in an llvm-objdump -d listing of the gfx950 code object.
usage: isa_loops.py all.s <kernel-name-substring> [min_loop_len]"""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    ins, base, on = [], None, False
    for l in open(path):
        m = re.match(r'^([0-9a-f]+) <(\S+)>:', l)
        if m:
            on = kern in m.group(2)
            if on:
                base = int(m.group(1), 16)
            continue
        if not on:
            continue
        m = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):\s*[0-9A-F ]+(?:<\S+\+0x([0-9a-f]+)>)?', l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), int(m.group(4), 16) + base if m.group(4) else None))
    idx = {a: i for i, (a, _, _, _) in enumerate(ins)}
    print(len(ins), "instructions in", kern)
    loops = sorted({(idx[t], i) for i, (a, op, _, t) in enumerate(ins) if t is not None and t <= a and t in idx})
    for s, e in loops:
        if e - s + 1 < min_len:
            continue
        cat = collections.Counter()
        for a, op, args, _ in ins[s:e + 1]:
            if op.startswith('v_'):
                k = 'VALU'
            elif op.startswith('ds_'):
                k = 'LDS'
            elif op.split('_')[0] in ('global', 'buffer', 'scratch', 'flat'):
                k = 'VMEM:' + op.split('_')[0]
            elif op.startswith('s_waitcnt'):
                k = 'waitcnt'
            elif op.startswith('s_nop'):
                k = 's_nop'
            elif op.startswith('s_'):
                k = 'SALU'
            else:
                k = op
            cat[k] += 1
        print("loop [%d..%d] len %d  @0x%x" % (s, e, e - s + 1, ins[s][0]), dict(cat))
        print("  ", collections.Counter(op for _, op, _, _ in ins[s:e + 1]).most_common(50))


main()
