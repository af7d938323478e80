"""Per-kernel SQ counter table of one eager training step.

Input: the counter_collection.csv rocprofv3 writes for
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
            SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d DIR -- \
            python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-secondary --no-roofline
Output: a markdown table on stdout (percentages of the kernel's summed wavefront cycles).
"""
import collections
import csv
import re
import sys


def main(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], name)
        if key not in seen:
            seen.add(key)
            calls[name] += 1
    print("| kernel | calls | wave cycles (M) | wait % | stall % | active % | VALU % | LDS % | VALU / wave |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc <= 0:
            continue
        pct = lambda k: 100.0 * c.get(k, 0) / wc
        waves = max(c.get("SQ_WAVES", 1), 1)
        print(f"| `{name}` | {calls[name]} | {wc / 1e6:.1f} | {pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | "
              f"{pct('SQ_ACTIVE_INST_ANY'):.1f} | {pct('SQ_ACTIVE_INST_VALU'):.1f} | {pct('SQ_ACTIVE_INST_LDS'):.1f} | "
              f"{c.get('SQ_INSTS_VALU', 0) / waves:.0f} |")


if __name__ == "__main__":
    main(sys.argv[1])
