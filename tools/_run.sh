set -o pipefail
mkdir -p gpurun_out/r5a
timeout -k 10 900 python -m pytest tests/test_gpu_forces.py tests/test_gpu_pass3_coef.py tests/test_gpu_ttm.py tests/test_gpu_scale_parity.py -x -q -m gpu -k "dg or distinct or golden or oracle or coef or stages or md_trajectory or both_kernel" > gpurun_out/r5a/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5a/tests.log
tail -5 gpurun_out/r5a/tests.log
for cfg in "shared:" "distinct:--tables distinct" "types2:--types 2"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 --warmup 30 $fl > gpurun_out/r5a/b_$name.json 2> gpurun_out/r5a/b_$name.err; echo "bench $name rc=$?"
  python - gpurun_out/r5a/b_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("%.1f steps/s %.3f ms" % (d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["kernel_ms_per_step"].items()}, d["roofline"]["kernel"][:60], d["config"]["kernel_family"])
except Exception as e: print("ERR", e)
PY
done
