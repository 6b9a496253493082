set -o pipefail
mkdir -p gpurun_out/r5k
for cfg in "n1:--no-pass1-ahead" "a1:" "n2:--no-pass1-ahead" "a2:" "n3:--no-pass1-ahead" "a3:"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline $fl > gpurun_out/r5k/b_$name.json 2> gpurun_out/r5k/b_$name.err; echo "bench $name rc=$?"
  python - gpurun_out/r5k/b_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("  %.1f steps/s %.3f ms" % (d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["kernel_ms_per_step"].items() if k in ("pass1","pass3","integrate","ttm_diff")})
except Exception as e: print("ERR", e)
PY
done
