#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: bench line + rocprofv3 kernel stats + HBM PMC passes.
# Outputs land in gpurun_out/<tag>/ ; tools/summarize_profiles.py reduces them into profiles/.
#   usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail   # (no -e: a rocprofv3 that does not exit after writing its database is killed by its timeout and the run goes on)
TAG=${1:-r01}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "[1/4] bench" | tee "$OUT/progress.log"
timeout -k 10 300 python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
echo "[2/4] rocprofv3 --kernel-trace --stats (same command)" | tee -a "$OUT/progress.log"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "[3/4] rocprofv3 --pmc FETCH_SIZE" | tee -a "$OUT/progress.log"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o run -- python3 "$ROOT/bench.py" "$@" --steps 10 --warmup 5 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
echo "[4/4] rocprofv3 --pmc WRITE_SIZE" | tee -a "$OUT/progress.log"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o run -- python3 "$ROOT/bench.py" "$@" --steps 10 --warmup 5 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
cd "$ROOT"
# keep only the small artefacts (raw traces are large): reduce now
python3 tools/summarize_profiles.py "$OUT" "$OUT/summary"
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
find "$OUT" -name '*_counter_collection.csv' -size +8M -delete
echo done | tee -a "$OUT/progress.log"
