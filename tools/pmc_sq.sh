#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the force kernels (two passes of <= 8 SQ counters), no TTM, 3 steps.
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-pmc_sq}; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  -d "$OUT/a" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-ttm --steps 3 --warmup 2 > "$OUT/a.json" 2> "$OUT/a.err"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAVES \
  -d "$OUT/b" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-ttm --steps 3 --warmup 2 > "$OUT/b.json" 2> "$OUT/b.err"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys, json
out = {}
for f in glob.glob(sys.argv[1] + "/*/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    for k, n, cnt, tot in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        if "tile" in k or "k_pass" in k:
            out.setdefault(k.split("(")[0], {})[n] = tot / cnt
json.dump(out, open(sys.argv[1] + "/sq_summary.json", "w"), indent=1)
for k, v in out.items():
    print(k)
    for n, x in sorted(v.items()): print("   %-26s %.4g" % (n, x))
PY
find "$OUT" -name '*.db' -size +4M -delete
