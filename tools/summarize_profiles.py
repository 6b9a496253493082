#!/usr/bin/env python3
"""Reduce the rocprofv3 outputs of tools/profile_bench.sh into small committed summaries.

  summarize_profiles.py <gpurun_out/tag> <dest-dir>

writes  kernel_stats.csv   (the rocprofv3 --stats table, verbatim, kernels only)
        hbm_traffic.json   per kernel: launches, FETCH_SIZE / WRITE_SIZE per launch as reported (KiB -> bytes),
                           and the gfx950-corrected figure (FETCH_SIZE x2 for wide coalesced reads, see
                           MI355X_MICROARCH.md "HBM": FETCH_SIZE tallies 128-B requests at 64 B).
"""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def counter_per_kernel(d, name):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):      # rocpd (ROCm 7.2 default output)
        con = sqlite3.connect(f)
        for kname, n, tot in con.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                         "where counter_name = ? group by kernel_name", (name,)):
            acc[kname][0] += n
            acc[kname][1] += tot
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != name:
                    continue
                a = acc[row["Kernel_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)
    dbs = glob.glob(os.path.join(src, "stats", "**", "*.db"), recursive=True)
    if stats:
        big = max(stats, key=os.path.getsize)
        open(os.path.join(dst, "kernel_stats.csv"), "w").write(open(big).read())
    elif dbs:
        con = sqlite3.connect(max(dbs, key=os.path.getsize))
        rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                           "from kernels group by name order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as fh:
            w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / total, 2), r[4], r[5]])
    fetch = counter_per_kernel(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = counter_per_kernel(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    out = {"_note": "FETCH_SIZE/WRITE_SIZE are reported in KiB by rocprofv3; per-launch averages over all dispatches of "
                    "the kernel in the run (bench.py --steps 10 --warmup 5).  fetch_bytes_corrected = 2 x FETCH_SIZE "
                    "(gfx950: 128-B read requests tallied at 64 B).  hbm_bytes = fetch_bytes_corrected + write_bytes."}
    for k in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        fb = vf / nf * 1024.0 if nf else None
        wb = vw / nw * 1024.0 if nw else None
        out[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_bytes_reported": fb,
                  "fetch_bytes_corrected": 2.0 * fb if fb is not None else None, "write_bytes": wb,
                  "hbm_bytes": (2.0 * fb if fb is not None else 0.0) + (wb or 0.0)}
    try:
        line = [l for l in open(os.path.join(src, "pmc_fetch.json")) if l.startswith("{")][-1]
        cfg = json.loads(line)["config"]
        out["_workload"] = cfg["workload"]
        out["_atoms_per_gpu"] = cfg["atoms_per_gpu"]
        out["_kernel_rev"] = json.loads(line)["roofline"].get("kernel_rev")      # bench.py refuses figures of another revision
    except Exception as e:                                  # keep the counters even if the bench line is missing
        out["_workload_error"] = str(e)
    json.dump(out, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    for f in ("bench.json", "pmc_fetch.json", "pmc_write.json", "bench_under_rocprof.json"):
        p = os.path.join(src, f)
        if os.path.exists(p):
            open(os.path.join(dst, f), "w").write(open(p).read())


if __name__ == "__main__":
    main()
