set -o pipefail
mkdir -p gpurun_out/r05
timeout -k 10 400 python bench.py > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_final.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_unit'] and {k: d['roofline_unit'][k] for k in ('lds','valu','hbm','source')}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
