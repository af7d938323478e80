#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): when do the preparation kernels (grid_query_kernel<16>) run relative to the others?"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id"), r.get("Stream_Id", "")) for r in rows]
ev.sort()
t_last = ev[-1][1]
ev = [e for e in ev if e[0] > t_last - 60_000_000]          # the last 60 ms: steady state
q = [e for e in ev if "grid_query_kernel<16>" in e[2]]
print("queues / streams seen:", sorted({(e[3], e[4]) for e in ev}))
for g in q[-6:]:
    over = [e for e in ev if e is not g and e[0] < g[1] and e[1] > g[0]]
    names = sorted({e[2].split("(")[0][:40] for e in over})
    print(f"grid_query<16> {(g[1] - g[0]) / 1e3:7.1f} us on queue {g[3]} stream {g[4]}: {len(over)} other dispatches overlap it: {names[:6]}")
# busy time vs wall time
span = ev[-1][1] - ev[0][0]
busy = sum(e[1] - e[0] for e in ev)
print(f"window {span / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms")
