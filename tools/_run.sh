set -o pipefail
mkdir -p gpurun_out/r5o
for cfg in "head:" "slab:--slab" "types2:--types 2" "dg:--tables distinct" "box1M:--box 64 64 64" "box40:--box 100 120 90 --steps 100"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline $fl > gpurun_out/r5o/b_$name.json 2> gpurun_out/r5o/b_$name.err; echo "bench $name rc=$?"
  python - gpurun_out/r5o/b_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("  %.1f steps/s %.3f ms" % (d["value"], d["ms_per_step"]), d["config"]["tile"], d["config"]["cells"], {k: round(v,3) for k,v in d["kernel_ms_per_step"].items() if k in ("pass1","pass3","nblist","ttm_diff")})
except Exception as e: print("ERR", e)
PY
done
