#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the force kernels (two passes of <= 8 SQ counters + GRBM_GUI_ACTIVE for the clock), 3 steps.
#   usage: tools/pmc_sq.sh <tag> [bench args...]      (default bench args: --no-ttm; the passes alone on the chip)
# Writes gpurun_out/<tag>/sq_summary.json: per kernel the per-launch averages of every counter, the average launch duration of the
# SAME runs (kernel trace) and, from them, what bench.py prints as `roofline_unit`:
#   lds  = SQ_LDS_IDX_ACTIVE / (CUs x cycles)            LDS-array cycles, bank conflicts included, over the launch's cycles
#   valu = 4 x SQ_ACTIVE_INST_VALU / (4 CUs x cycles)    quad-cycles a SIMD's vector ALU was executing, over the SIMDs' cycles
#   cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs), i.e. the clock the chip held during the launch x its duration
set -e -o pipefail
TAG=${1:-pmc_sq}; shift || true
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(--no-ttm)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
  -d "$OUT/a" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 2 "${ARGS[@]}" > "$OUT/a.json" 2> "$OUT/a.err"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE \
  -d "$OUT/b" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 2 "${ARGS[@]}" > "$OUT/b.json" 2> "$OUT/b.err"
cd "$ROOT"
python3 tools/sq_summary.py "$OUT"
find "$OUT" -name '*.db' -size +4M -delete
