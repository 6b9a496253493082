set -o pipefail
R=gpurun_out/r5p
mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_gpu_ttm.py tests/test_gpu_ttm_persist.py tests/test_gpu_scale_parity.py tests/test_gpu_fdtd.py -x -q -m gpu > $R/tests.log 2>&1; echo "tests rc=$?" | tee -a $R/tests.log
tail -4 $R/tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline --box 272 272 272 --slab --steps 20 --warmup 5 > $R/b_80M.json 2> $R/b_80M.err; echo "80M rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > $R/b_head.json 2> $R/b_head.err; echo "head rc=$?"
for n in 80M head; do python - $R/b_$n.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("  %.1f steps/s %.3f ms" % (d["value"], d["ms_per_step"]), d["config"]["atoms_per_gpu"], {k: round(v,3) for k,v in d["kernel_ms_per_step"].items() if k in ("pass1","pass3","ttm_diff","nblist","integrate")}, d["roofline"]["kernel"][:40], "|", d["roofline_other"]["kernel"][:40])
except Exception as e: print("ERR", e)
PY
done
