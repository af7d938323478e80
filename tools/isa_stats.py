#!/usr/bin/env python3
"""Per-kernel facts from a hipcc -S listing: registers, LDS, scratch, instruction counts of interest.
usage: python tools/isa_stats.py file.s substring"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_Z\S*' + re.escape(pat) + r'\S*):', s, re.M):
    name = m.group(1)
    start = m.end()
    end = s.index('.Lfunc_end', start)
    body = s[start:end]
    def sym(k):
        r = re.search(r'\.set ' + re.escape(name) + r'\.' + k + r', (\d+)', s)
        return r.group(1) if r else '?'
    lds = re.search(r'\.amdhsa_kernel ' + re.escape(name) + r'\n(.*?)\.end_amdhsa_kernel', s, re.S)
    l = re.search(r'group_segment_fixed_size (\d+)', lds.group(1)).group(1) if lds else '?'
    cnt = lambda rx: len(re.findall(rx, body))
    print(name[:90])
    v0, vn = cnt(r'vmcnt\(0\)'), cnt(r'vmcnt\([1-9]')
    print("  vgpr %s sgpr %s scratch %s lds %s | glds %d mfma %d ds_read %d vmcnt(0) %d vmcnt(n) %d s_barrier %d lines %d" % (
        sym('num_vgpr'), sym('numbered_sgpr'), sym('private_seg_size'), l, cnt(r'global_load_lds'), cnt(r'v_mfma'),
        cnt(r'ds_read'), v0, vn, cnt(r's_barrier'), body.count(chr(10))))
