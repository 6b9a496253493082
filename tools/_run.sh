set -o pipefail
mkdir -p gpurun_out/r5c
timeout -k 10 900 python -m pytest tests/test_gpu_forces.py tests/test_gpu_ttm.py -x -q -m gpu -k "golden or both_kernel or two_types or trajectory_vs_reference or 2types" > gpurun_out/r5c/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5c/tests.log
tail -3 gpurun_out/r5c/tests.log
for cfg in "types2:--types 2"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 --warmup 30 $fl > gpurun_out/r5c/b_$name.json 2> gpurun_out/r5c/b_$name.err; echo "bench $name rc=$?"
  python - gpurun_out/r5c/b_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("%.1f steps/s %.3f ms" % (d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["kernel_ms_per_step"].items()}, d["roofline"]["kernel"][:60], d["config"]["kernel_family"])
except Exception as e: print("ERR", e)
PY
done
