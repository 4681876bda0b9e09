"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see scripts/gpu_prof.sh).

usage: python scripts/pmc_traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Counter units are KiB-like "KB" as rocprofv3 reports them.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies
128-byte requests of wide coalesced reads at 64 B, so fetch bytes = 2 x FETCH_SIZE; WRITE_SIZE is taken as reported (it matches the
known output sizes of the write-once kernels here, e.g. the stem).  Rows are keyed by (kernel, workgroups) because one kernel serves
several problem shapes; values are averages per launch.
"""
import csv
import json
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import demangle  # noqa: E402


def short(n):
    n = demangle(n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.replace("fvit::", "")
    n = n.split("(")[0]
    return n.replace("_Float16", "f16").replace("__bf16", "bf16")


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            a = acc[key]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for key, a in fetch.items():
        w = write.get(key)
        if not w or "fvit" not in key[0] and "kernel" not in key[0]:
            continue
        fetch_mb = 2.0 * a[1] / a[0] * 1024 / 1e6
        write_mb = w[1] / w[0] * 1024 / 1e6
        rows.append({"kernel": key[0], "workgroups": key[1], "launches": a[0], "avg_us_under_pmc": round(a[2] / a[0], 2),
                     "fetch_size_raw_kb": round(a[1] / a[0], 1), "write_size_raw_kb": round(w[1] / w[0], 1),
                     "hbm_read_mb": round(fetch_mb, 3), "hbm_write_mb": round(write_mb, 3), "hbm_traffic_mb": round(fetch_mb + write_mb, 3),
                     "total_us": round(a[2], 1)})
    rows.sort(key=lambda r: -r["total_us"])
    json.dump({"note": "HBM bytes per launch from rocprofv3 PMC passes of `bench.py --no-graph` (eager, default stream shards); "
                       "read = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE; units MB = 1e6 bytes",
               "kernels": rows}, open(sys.argv[3], "w"), indent=1)
    print(f"{len(rows)} (kernel, grid) rows -> {sys.argv[3]}")
    for r in rows[:12]:
        print(r["kernel"][:60], r["workgroups"], r["hbm_read_mb"], r["hbm_write_mb"])


if __name__ == "__main__":
    main()
