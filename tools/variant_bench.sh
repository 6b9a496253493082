#!/bin/bash
# Run ON THE GPU BOX (via gpurun): build a kernel variant of libimdgpu.so outside the tree and bench it.
#   usage: tools/variant_bench.sh <name> "<EXTRA make flags>" "<env assignments>" [bench args...]
#   (experiment switches of the environment need -DIMDGPU_EXPERIMENTS among the EXTRA flags; bench.py is told --allow-experiments)
# Results: gpurun_out/variants/<name>.json (the bench line) ; the in-tree library is not touched.
set -e -o pipefail
NAME=$1; EXTRA=$2; ENVS=$3; shift 3
ROOT=$(pwd)
W=/tmp/imdgpu_variant_$NAME
rm -rf "$W"; mkdir -p "$W/imd_amd" "$ROOT/gpurun_out/variants"
cp -r "$ROOT/imd_amd/csrc" "$W/imd_amd/csrc"; cp -r "$ROOT/include" "$W/include"
make -s -C "$W/imd_amd/csrc" clean
make -s -C "$W/imd_amd/csrc" EXTRA="$EXTRA" libimdgpu.so 2> "$ROOT/gpurun_out/variants/$NAME.build.log"
env $ENVS IMDGPU_LIB="$W/imd_amd/csrc/libimdgpu.so" timeout -k 10 400 python3 bench.py --no-cpu-baseline --allow-experiments "$@" \
    > "$ROOT/gpurun_out/variants/$NAME.json" 2> "$ROOT/gpurun_out/variants/$NAME.err"
python3 - "$ROOT/gpurun_out/variants/$NAME.json" "$NAME" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernel_ms_per_step"]; t = d["config"]["tile_info"]
print("%-28s %.3f ms/step  pass1 %.3f pass3 %.3f nbl %.3f  tile %s threads %d lds1 %d slots %d" % (sys.argv[2], d["ms_per_step"],
      k.get("pass1", 0), k.get("pass3", 0), k.get("nblist", 0), d["config"]["tile"], t["tile_threads"], t["lds_pass1"], t["slot_cap"]))
PY
