set -o pipefail
R=gpurun_out/r05
mkdir -p $R
bash tools/profile_bench.sh r05/headline > $R/headline.log 2>&1; echo "profile rc=$?"; tail -2 $R/headline.log
bash tools/pmc_sq.sh r05/sq_ttm --temp 0.0258 > $R/sq_ttm.log 2>&1; echo "sq ttm rc=$?"; tail -7 $R/sq_ttm.log
bash tools/pmc_sq.sh r05/sq_nottm --no-ttm > $R/sq_nottm.log 2>&1; echo "sq nottm rc=$?"
bash tools/pmc_sq.sh r05/sq_dg --no-ttm --tables distinct > $R/sq_dg.log 2>&1; echo "sq dg rc=$?"
bash tools/pmc_sq.sh r05/sq_types2 --no-ttm --types 2 > $R/sq_types2.log 2>&1; echo "sq t2 rc=$?"
timeout -k 10 600 python bench.py --no-cpu-baseline --box 272 272 272 --slab --steps 20 --warmup 5 > $R/b_80M.json 2> $R/b_80M.err; echo "80M rc=$?"
timeout -k 10 500 python bench.py --no-cpu-baseline --steps 3000 --warmup 50 > $R/b_3000.json 2> $R/b_3000.err; echo "3000 steps rc=$?"
for cfg in "slab:--slab" "fdtd:--laser fdtd --steps 60 --warmup 120" "fdtd_dl:--laser fdtd --dl-tables --steps 60 --warmup 120" "distinct:--tables distinct" "types2:--types 2" "types2c:--types 2 --types2-rows 0.3" "nottm:--no-ttm" "rccl_dec:--force-distributed --init-dist --backend nccl --ttm-lattice decomposed" "rccl_rep:--force-distributed --init-dist --backend nccl --ttm-lattice replicated"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  IMDGPU_DIST_SELF_COMM=1 timeout -k 10 300 python bench.py --no-cpu-baseline $fl > $R/b_$name.json 2> $R/b_$name.err; echo "bench $name rc=$?"
  python - $R/b_$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("  %.1f steps/s %.3f ms" % (d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["kernel_ms_per_step"].items()}, d["rebuild_ms"]["make_nblist_ms_each"], d["rebuild_ms"]["fix_cells_ms_each"])
except Exception as e: print("ERR", e)
PY
done
