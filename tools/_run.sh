set -o pipefail
mkdir -p gpurun_out/r5g
timeout -k 10 900 python -m pytest tests/test_gpu_ttm.py tests/test_gpu_ttm_persist.py tests/test_gpu_scale_parity.py -x -q -m gpu > gpurun_out/r5g/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5g/tests.log
tail -4 gpurun_out/r5g/tests.log
for cfg in "shared:" "slab:--slab"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline $fl > gpurun_out/r5g/b_$name.json 2> gpurun_out/r5g/b_$name.err; echo "bench $name rc=$?"
  python - gpurun_out/r5g/b_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("%.1f steps/s %.3f ms" % (d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["kernel_ms_per_step"].items()})
except Exception as e: print("ERR", e)
PY
done
