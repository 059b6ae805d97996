#!/usr/bin/env python3
"""Development helper: the order of matrix instructions in the largest basic block of a kernel ('4' = 4x4x4, 'V' = 16x16x32, 'W' = 32x32x16)
with the instruction index of each - shows whether a hand-written interleave survived the compiler.
    python tools/isa_mfma_order.py file.s kernel-name-substring"""
import re
import sys
s = open(sys.argv[1]).read()
for name in [n for n in re.findall(r'^(_Z\w+):\s*(?:;.*)?$', s, re.M) if sys.argv[2] in n]:
    tail = s[s.index(name + ':'):]
    end = re.search(r'^\.Lfunc_end\d+:', tail, re.M)
    body = tail[:end.start()]
    blocks = re.split(r'^(\.LBB\d+_\d+):.*$', body, flags=re.M)
    best = None
    for i in range(1, len(blocks), 2):
        lines = [l for l in blocks[i + 1].splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
        if best is None or len(lines) > len(best[1]):
            best = (blocks[i], lines)
    seq = [(i, '4' if '4x4x4' in l else 'V' if '16x16x32' in l else 'W') for i, l in enumerate(best[1]) if 'v_mfma' in l]
    print(name, best[0], len(best[1]), 'instructions')
    print('  ', ''.join(k for _, k in seq))
    print('  ', [i for i, _ in seq])
