#!/usr/bin/env python3
"""Instruction mix of the largest loop of a kernel in a hipcc -S listing.  usage: python tools/isa_loop_mix.py file.s name-substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for name in re.findall(r'^(_Z\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M):
    i = s.index(name + ':')
    lines = s[i:s.index('.Lfunc_end', i)].split('\n')
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    best = None
    for n, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            span = n - labels[m.group(1)]
            if best is None or span > best[0]:
                best = (span, labels[m.group(1)], n)
    loop = lines[best[1]:best[2] + 1]
    ins = [l.strip().split()[0] for l in loop if l.startswith('\t') and not l.strip().startswith((';', '.'))]
    c = Counter()
    for x in ins:
        k = ('s_cbranch' if x.startswith('s_cbranch') else 's_waitcnt' if x.startswith('s_waitcnt') else 'scalar' if x.startswith('s_')
             else 'mfma' if x.startswith('v_mfma') else 'v_cvt' if x.startswith('v_cvt') else 'valu' if x.startswith('v_')
             else 'lds' if x.startswith('ds_') else 'vmem' if x.startswith(('global_', 'buffer_')) else x)
        c[k] += 1
    print(name[:100]); print('  loop instructions', len(ins), dict(c))
