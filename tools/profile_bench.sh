#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: bench line + rocprofv3 kernel stats + HBM PMC passes.
# Outputs land in gpurun_out/<tag>/ ; tools/summarize_profiles.py reduces them into profiles/.
#   usage: tools/profile_bench.sh <tag> [bench args...]
# Every step's exit code is checked and written to <tag>/status.txt; a step that fails or runs into its timeout (a backstop, 124 / 137)
# FAILS the script after the remaining steps have run -- partial data never reaches the summary unnoticed.
set -o pipefail
TAG=${1:-r01}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/status.txt"
FAILED=0
step() {   # step <name> <stdout file> <stderr file> <command...>
  local name=$1 so=$2 se=$3; shift 3
  echo "[$name]" | tee -a "$OUT/progress.log"
  "$@" > "$so" 2> "$se"; local rc=$?
  echo "$name rc=$rc" | tee -a "$OUT/status.txt"
  if [ $rc -ne 0 ]; then FAILED=1; echo "  step $name FAILED (rc $rc; 124/137 = timeout): tail of $se" ; tail -5 "$se"; fi
  return 0
}
step bench "$OUT/bench.json" "$OUT/bench.err" timeout -k 10 300 python3 bench.py "$@"
cat "$OUT/bench.json"
cd /tmp
step stats "$OUT/bench_under_rocprof.json" "$OUT/stats.err" timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline
step pmc_fetch "$OUT/pmc_fetch.json" "$OUT/pmc_fetch.err" timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o run -- python3 "$ROOT/bench.py" "$@" --steps 10 --warmup 5 --no-cpu-baseline
step pmc_write "$OUT/pmc_write.json" "$OUT/pmc_write.err" timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o run -- python3 "$ROOT/bench.py" "$@" --steps 10 --warmup 5 --no-cpu-baseline
cd "$ROOT"
# keep only the small artefacts (raw traces are large): reduce now
if [ $FAILED -eq 0 ]; then
  python3 tools/summarize_profiles.py "$OUT" "$OUT/summary" || FAILED=1
fi
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
find "$OUT" -name '*_counter_collection.csv' -size +8M -delete
echo "done failed=$FAILED" | tee -a "$OUT/progress.log"
exit $FAILED
