#!/bin/bash
# Run ON THE GPU BOX: bench prebuilt kernel variants (build/variants/<name>/imd_amd/csrc/libimdgpu.so, cross-compiled in the
# dev container) one after the other.  usage: tools/variant_run.sh "<bench args>" name[:ENV=V,...] ...   ("tree" = the in-tree library)
ARGS=$1; shift
mkdir -p gpurun_out/variants
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  lib=""; [ "$name" != "tree" ] && lib="IMDGPU_LIB=$PWD/build/variants/${name%%@*}/imd_amd/csrc/libimdgpu.so"
  tag=$(echo "$spec" | tr ':=,@' '____')
  env $envs $lib timeout -k 10 400 python3 bench.py --no-cpu-baseline $ARGS > gpurun_out/variants/$tag.json 2> gpurun_out/variants/$tag.err || echo "$tag FAILED"
  python3 - gpurun_out/variants/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step"]; t = d["config"]["tile_info"]
    print("%-34s %.3f ms/step  pass1 %.3f pass3 %.3f nbl %.3f ttm %.3f ufd %.3f  tile %s" % (sys.argv[2], d["ms_per_step"], k.get("pass1", 0), k.get("pass3", 0),
          k.get("nblist", 0), k.get("ttm_diff", 0), k.get("ttm_update_fd", 0), d["config"]["tile"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
