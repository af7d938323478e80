#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a per-kernel PMC table (tools/pmc_hbm_table.py): HBM-side bytes per LAUNCH keyed
"<function>|<op>" - the op string bench.py prints for a launch shape - plus "<function>|*" = bytes per STEP over all launches of
the function.  The instantiations of vpool_bwd_kernel<DT, TERMS, SRC, NW, GB, ACC> map to ops by DT (d = 16 DT; both stages of a
level share one op string: their mean):   python tools/pmc_traffic_json.py profiles/r05_pmc_hbm_step.md [points_level0=327680]"""
import json
import os
import re
import sys

md = sys.argv[1]
P0 = int(sys.argv[2]) if len(sys.argv) > 2 else 327680
rows = {}
for line in open(md):
    m = re.match(r"\| `([^`]+)` \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if m:
        rows[m.group(1)] = (float(m.group(2)), float(m.group(5)) * 1e6)       # launches per step, bytes per step
out = {}
for fn in ("vpool_bwd_kernel", "vpool_fwd_kernel"):
    per_dt, total = {}, 0.0
    for name, (launches, nbytes) in rows.items():
        if name.startswith(fn + "<"):
            dt = int(name[len(fn) + 1:].split(",")[0])
            per_dt.setdefault(dt, []).append(nbytes / launches)
            total += nbytes
    cat = "pool_bwd" if "bwd" in fn else "pool_fwd_virtual"
    for dt, v in per_dt.items():
        pts = P0 // (4 ** {1: 0, 2: 1, 4: 1}[dt]) if dt != 2 else P0 // 4
        out[f"{fn}|{cat}[{pts}, 16, {16 * dt}]"] = int(sum(v) / len(v))
    if total:
        out[f"{fn}|*"] = int(total)
for fn in ("segment_sum_vec_kernel", "wgemm2_kernel", "sgemm_kernel", "pwgrad128w_batch_kernel", "bn_bwd_apply_vec_kernel",
           "bn_bwd_reduce_vec_kernel", "pool128_bwd_kernel", "swgrad_batch_kernel"):
    tot = sum(b for n, (_, b) in rows.items() if n.split("<")[0] == fn)
    if tot:
        out[f"{fn}|*"] = int(tot)
path = os.path.join(os.path.dirname(os.path.abspath(md)), "pmc_traffic.json")
old = json.load(open(path)) if os.path.exists(path) else {}
for k, v in old.items():          # the wide GEMM's single-shape measurements of rounds 2 / 3 (r03_pmc_wgemm2_*.md) stay, re-keyed
    if k.startswith("gemm["):
        out[f"wgemm2_kernel|{k}"] = v
out["_note"] = (f"HBM-side bytes, FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 correction of the guide; gathered rows: upper bound), from {os.path.basename(md)}: "
                "'<function>|<op>' = per launch of that shape (both pooling stages of a level share an op: their mean), '<function>|*' = per step over "
                "all launches of the function; wgemm2_kernel|gemm[...] from r03_pmc_wgemm2_*.md")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
