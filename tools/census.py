#!/usr/bin/env python3
"""Per-function launches / us per step from a rocprofv3 kernel_stats.csv:  python tools/census.py CSV steps [other CSV steps]"""
import csv, re, sys
def load(path, steps):
    agg = {}
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "").split("(")[0]
        f = n.split("<")[0]
        a = agg.setdefault(f, [0, 0.0])
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
    return {f: (c / steps, t / steps / 1e3) for f, (c, t) in agg.items()}
a = load(sys.argv[1], float(sys.argv[2]))
b = load(sys.argv[3], float(sys.argv[4])) if len(sys.argv) > 4 else {}
tot = [0, 0, 0, 0]
for f in sorted(set(a) | set(b), key=lambda f: -(a.get(f, (0, 0))[1])):
    x, y = a.get(f, (0, 0)), b.get(f, (0, 0))
    tot[0] += x[0]; tot[1] += x[1]; tot[2] += y[0]; tot[3] += y[1]
    print(f"{f[:40]:40s} {x[0]:6.1f} launches {x[1]:8.1f} us" + (f"   | {y[0]:6.1f} {y[1]:8.1f} us  d {x[1] - y[1]:+7.1f}" if b else ""))
print(f"{'TOTAL':40s} {tot[0]:6.1f} launches {tot[1]:8.1f} us" + (f"   | {tot[2]:6.1f} {tot[3]:8.1f} us" if b else ""))
