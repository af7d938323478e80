"""Per-kernel HBM traffic of one eager training step from two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot share one):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -- python3 bench.py --steps 2 --warmup 1 --no-graph \
      --no-cpu-baseline --no-inference --no-secondary --no-roofline --no-other-configs
  rocprofv3 --pmc WRITE_SIZE ... -d DIR_W -- (same)
  python tools/pmc_hbm_table.py DIR_F DIR_W STEPS        (STEPS = 0: counted per pass from the adam_kernel launches)
gfx950 corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024
(the factor 2 is calibrated for wide streaming reads; narrow / gathered reads are uncalibrated - read them as an upper bound).
Output: markdown table per kernel function, per step."""
import collections
import csv
import glob
import re
import sys


def load(d, counter):
    acc, calls, dur = collections.defaultdict(float), collections.Counter(), collections.defaultdict(float)
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            acc[name] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], name)
            if key not in seen:
                seen.add(key)
                calls[name] += 1
    for path in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            dur[name] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, calls, dur


def main(df, dw, steps):
    f, calls, dur = load(df, "FETCH_SIZE")
    w, wcalls, _ = load(dw, "WRITE_SIZE")
    # the warm-up of bench.py is time-based: the two passes need not hold the same number of steps - adam_kernel runs once per step
    steps_w = steps
    if steps <= 0:
        steps, steps_w = float(calls["adam_kernel"]), float(wcalls["adam_kernel"])
    rows = []
    for k in set(f) | set(w):
        rd, wr = f.get(k, 0.0) * 1024 * 2 / steps, w.get(k, 0.0) * 1024 / steps_w
        rows.append((rd + wr, k, calls[k] / steps, rd, wr, dur.get(k, 0.0) / steps))
    tot = sum(r[0] for r in rows)
    print(f"HBM-side bytes per step: {tot / 1e9:.2f} GB (read {sum(r[3] for r in rows) / 1e9:.2f}, write {sum(r[4] for r in rows) / 1e9:.2f})\n")
    print("| kernel | launches/step | read MB | write MB | total MB | share | us/step (under the profiler) | GB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for t, k, c, rd, wr, d in sorted(rows, reverse=True):
        if t < 1e6:
            continue
        print(f"| `{k}` | {c:.1f} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | {t / 1e6:.1f} | {100 * t / tot:.1f} % | {d / 1e3:.0f} | {t / max(d, 1):.0f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]))
