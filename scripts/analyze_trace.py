"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py: how well do the stream shards overlap?

usage: python scripts/analyze_trace.py <kernel_trace.csv> [window_ms]
Looks at the last `window_ms` (default 40) of the trace (the timed graph replays) and prints wall time, the union of kernel-busy
time, the time spent with k = 0, 1, 2, ... kernels in flight, and per kernel name: launches, summed duration, mean duration,
time during which it was the only kernel running.
"""
import csv
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import demangle  # noqa: E402

path = sys.argv[1]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 40e6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort()
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - win]
t0 = min(r[0] for r in rows)
wall = t_end - t0


def short(n):
    n = demangle(n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.replace("fvit::", "")
    return n.split("(")[0][:70]


ev = []
for i, (s, e, n, g) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort()
active = set()
hist = defaultdict(int)
solo = defaultdict(int)
prev = t0
for t, d, i in ev:
    dt = t - prev
    if dt > 0:
        hist[len(active)] += dt
        if len(active) == 1:
            solo[short(rows[next(iter(active))][2])] += dt
    prev = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
print(f"window {wall / 1e6:.3f} ms, {len(rows)} kernel launches, summed kernel time {sum(e - s for s, e, _, _ in rows) / 1e6:.3f} ms")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {hist[k] / 1e6:8.3f} ms  ({100.0 * hist[k] / wall:5.1f} %)")
agg = defaultdict(lambda: [0, 0, 0])
for s, e, n, g in rows:
    a = agg[(short(n), g)]
    a[0] += 1
    a[1] += e - s
print(f"{'kernel':70s} {'wgs':>6s} {'calls':>6s} {'sum ms':>8s} {'avg us':>8s} {'share':>6s}")
tot = sum(a[1] for a in agg.values())
for (n, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n[:70]:70s} {g:6d} {a[0]:6d} {a[1] / 1e6:8.3f} {a[1] / a[0] / 1e3:8.1f} {100.0 * a[1] / tot:5.1f}%")
print("solo time (only kernel in flight):")
for n, t in sorted(solo.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {n[:70]:70s} {t / 1e6:8.3f} ms")
