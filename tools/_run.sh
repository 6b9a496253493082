set -o pipefail
R=gpurun_out/r5q
mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_gpu_forces.py tests/test_gpu_scale_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_pass3_coef.py -x -q -m gpu > $R/tests.log 2>&1; echo "tests rc=$?" | tee -a $R/tests.log
tail -4 $R/tests.log
for cfg in "nottm:--no-ttm" "nottm_dg:--no-ttm --tables distinct" "c2:--no-ttm --box 64 64 64 --steps 400"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline $fl > $R/b_$name.json 2> $R/b_$name.err; echo "bench $name rc=$?"
  python - $R/b_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("  %.1f steps/s %.3f ms" % (d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["kernel_ms_per_step"].items() if k in ("pass1","pass3","nblist","integrate")})
except Exception as e: print("ERR", e)
PY
done
