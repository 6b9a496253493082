#!/usr/bin/env python3
"""Reduce the two rocprofv3 --pmc passes of tools/pmc_sq.sh into sq_summary.json (see the header of that script)."""
import glob
import json
import os
import sqlite3
import sys

NUM_CU = 256


def tables(con):
    return [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]


def main():
    out_dir = sys.argv[1]
    out = {}
    for sub in ("a", "b"):
        for f in glob.glob(os.path.join(out_dir, sub, "**", "*.db"), recursive=True):
            con = sqlite3.connect(f)
            for k, n, cnt, tot in con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                                              "group by kernel_name, counter_name"):
                if "tile" in k or "k_pass" in k:
                    e = out.setdefault(k.split("(")[0], {})
                    key = n if n != "GRBM_GUI_ACTIVE" else "GRBM_GUI_ACTIVE_" + sub
                    e[key] = tot / cnt
            try:      # average duration of the kernel's dispatches in the same run (ns)
                for k, cnt, avg in con.execute("select name, count(*), avg(duration) from kernels group by name"):
                    if k.split("(")[0] in out:
                        out[k.split("(")[0]]["avg_ns_" + sub] = avg
                        out[k.split("(")[0]]["launches_" + sub] = cnt
            except sqlite3.Error as e:
                out.setdefault("_errors", []).append("%s: %s (tables: %s)" % (f, e, tables(con)[:12]))
    for k, v in out.items():
        if not isinstance(v, dict):
            continue
        ca = v.get("GRBM_GUI_ACTIVE_a", 0) / 8.0
        cb = v.get("GRBM_GUI_ACTIVE_b", 0) / 8.0
        if ca and "SQ_ACTIVE_INST_VALU" in v:
            v["valu_busy"] = 4.0 * v["SQ_ACTIVE_INST_VALU"] / (4 * NUM_CU * ca)
            if v.get("avg_ns_a"):
                v["clock_ghz_a"] = ca / v["avg_ns_a"]
        if cb and "SQ_LDS_IDX_ACTIVE" in v:
            v["lds_busy"] = v["SQ_LDS_IDX_ACTIVE"] / (NUM_CU * cb)
            v["lds_conflict_share"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v["SQ_LDS_IDX_ACTIVE"], 1.0)
            if v.get("avg_ns_b"):
                v["clock_ghz_b"] = cb / v["avg_ns_b"]
    try:
        line = [l for l in open(os.path.join(out_dir, "a.json")) if l.startswith("{")][-1]
        d = json.loads(line)
        out["_workload"] = d["config"]["workload"]; out["_atoms_per_gpu"] = d["config"]["atoms_per_gpu"]
        out["_kernel_rev"] = d["roofline"].get("kernel_rev")
    except Exception as e:      # noqa: BLE001
        out["_workload_error"] = str(e)
    json.dump(out, open(os.path.join(out_dir, "sq_summary.json"), "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print(k, {q: round(v[q], 3) for q in ("valu_busy", "lds_busy", "lds_conflict_share", "clock_ghz_a", "clock_ghz_b") if q in v},
                  "avg_us", round(v.get("avg_ns_a", 0) / 1e3, 1))


if __name__ == "__main__":
    main()
