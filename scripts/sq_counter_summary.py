"""Per-(kernel, grid) SQ counter summary from one rocprofv3 --pmc pass (SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT) of `bench.py --no-graph`.

usage: python scripts/sq_counter_summary.py <counter_collection.csv> <out.json>

Derived per launch (averages): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel duration x clock) is not derivable
without the clock, so the file reports the RATIOS that are: wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked on s_waitcnt /
barrier), issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, and
mfma_busy_per_wave_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) (SQ_WAVE_CYCLES counts quad-cycles, MI355X_MICROARCH.md).
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic_summary import short  # noqa: E402  (re-uses the kernel-name shortening; its module-level code needs argv)


def main():
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(float)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[key][r["Counter_Name"]] += 1
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = []
    for key, c in acc.items():
        n = max(cnt[key].get("SQ_WAVE_CYCLES", 0), 1)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0 or ("kernel" not in key[0] and "fvit" not in key[0]):
            continue
        rows.append({"kernel": key[0], "workgroups": key[1], "launches": n, "avg_us_under_pmc": round(dur[key] / n, 2), "total_us": round(dur[key], 1),
                     "wait_frac": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3), "issue_stall_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                     "active_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
                     "mfma_busy_per_wave_cycle": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wc), 3),
                     "valu_insts_per_launch": round(c.get("SQ_INSTS_VALU", 0.0) / n, 0), "lds_bank_conflict_per_launch": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / n, 0),
                     "wave_quad_cycles_per_launch": round(wc / n, 0)})
    rows.sort(key=lambda r: -r["total_us"])
    json.dump({"note": "SQ counters per launch from one rocprofv3 PMC pass of `bench.py --no-graph` (eager, the default launch structure: 2 stream shards joined in front of level 3); ratios to SQ_WAVE_CYCLES",
               "kernels": rows}, open(sys.argv[2], "w"), indent=1)
    print(f"{len(rows)} (kernel, grid) rows -> {sys.argv[2]}")
    for r in rows[:10]:
        print(r["kernel"][:50], r["workgroups"], "wait", r["wait_frac"], "stall", r["issue_stall_frac"], "active", r["active_frac"], "mfma", r["mfma_busy_per_wave_cycle"])


if __name__ == "__main__":
    main()
