#!/usr/bin/env python3
"""Copy what tools/_profile_set.sh left under gpurun_out/<round>/ into profiles/ under the names the documents cite.
usage: tools/collect_profiles.py r05"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", rnd)
dst = os.path.join(ROOT, "profiles")
pairs = [
    ("headline/summary/bench.json", "bench_10M_ttm.json"),
    ("headline/summary/bench_under_rocprof.json", "bench_10M_ttm_under_rocprofv3.json"),
    ("headline/summary/kernel_stats.csv", "rocprofv3_kernel_stats_10M_ttm.csv"),
    ("headline/summary/hbm_traffic.json", "hbm_traffic_10M_ttm.json"),
    ("sq_ttm/sq_summary.json", "sq_counters_force_kernels_TTM.json"),
    ("sq_nottm/sq_summary.json", "sq_counters_force_kernels_noTTM.json"),
    ("sq_dg/sq_summary.json", "sq_counters_force_kernels_noTTM_distinct_grids.json"),
    ("sq_types2/sq_summary.json", "sq_counters_force_kernels_noTTM_two_types.json"),
    ("b_slab.json", "bench_10M_slab_ttm.json"),
    ("b_fdtd.json", "bench_10M_slab_fdtd.json"),
    ("b_fdtd_dl.json", "bench_10M_slab_fdtd_dl_tables.json"),
    ("b_distinct.json", "bench_10M_ttm_distinct_grids.json"),
    ("b_types2.json", "bench_10M_ttm_two_types.json"),
    ("b_nottm.json", "bench_10M_noTTM.json"),
    ("b_rccl_dec.json", "bench_10M_ttm_force_distributed_rccl_decomposed.json"),
    ("b_rccl_rep.json", "bench_10M_ttm_force_distributed_rccl_replicated.json"),
    ("b_80M.json", "bench_80M_slab_ttm_one_gpu.json"),
    ("b_3000.json", "bench_10M_ttm_3000_steps.json"),
    ("b_types2c.json", "bench_10M_ttm_two_types_coarse_tables.json"),
]
for a, b in pairs:
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copyfile(p, os.path.join(dst, "%s_%s" % (rnd, b)))
        print("%s -> profiles/%s_%s" % (a, rnd, b))
    else:
        print("missing:", a)
