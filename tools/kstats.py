#!/usr/bin/env python3
"""Per-kernel call count / average / minimum (us) from a rocprofv3 --stats directory:  python tools/kstats.py DIR [name-substring ...]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
keys = sys.argv[2:]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "").split("(")[0]
    if not keys or any(k in n for k in keys):
        print(f"{n[:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f}")
