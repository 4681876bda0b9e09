"""Summarise a rocprofv3 --kernel-trace --stats results database (rocpd sqlite) into the two CSVs kept under profiles/:
   <out>_kernel_stats.csv          per kernel name: calls, total/avg/min/max duration (ns), share
   <out>_fvit_kernels_by_shape.csv fvit kernels split by grid size (the same kernel serves several problem shapes)
usage: python scripts/summarize_rocprof_db.py <bench_results.db> <out-prefix>
"""
import csv
import sqlite3
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import demangle  # noqa: E402

db = sqlite3.connect(sys.argv[1])
out = sys.argv[2]
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select * from kernels").fetchall()
ix = {c: i for i, c in enumerate(cols)}
name_c = "name" if "name" in ix else "kernel_name"
by_name = defaultdict(list)
by_shape = defaultdict(list)
for r in rows:
    d = r[ix["end"]] - r[ix["start"]]
    n = demangle(r[ix[name_c]])
    by_name[n].append(d)
    wg = r[ix["workgroup_x"]] if "workgroup_x" in ix else r[ix["workgroup_size_x"]]
    gx = r[ix["grid_x"]] if "grid_x" in ix else r[ix["grid_size_x"]]
    if "fvit" in r[ix[name_c]] or "_kernel" in n:
        by_shape[(n, gx // max(wg, 1), wg, r[ix["vgpr_count"]] if "vgpr_count" in ix else 0,
                  r[ix["lds_size"]] if "lds_size" in ix else (r[ix["lds_block_size"]] if "lds_block_size" in ix else 0))].append(d)
tot = sum(sum(v) for v in by_name.values())
with open(out + "_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, v in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 3), min(v), max(v)])
with open(out + "_fvit_kernels_by_shape.csv", "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Workgroups", "WorkgroupSize", "VGPR", "LDS", "Calls", "AverageUs", "TotalUs"])
    for (n, g, wg, vg, lds), v in sorted(by_shape.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, g, wg, vg, lds, len(v), round(sum(v) / len(v) / 1e3, 3), round(sum(v) / 1e3, 3)])
print("kernels:", len(rows), "distinct:", len(by_name), "total ms:", tot / 1e6)
