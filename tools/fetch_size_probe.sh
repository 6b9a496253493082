#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of tools/fetch_size_probe.hip -> gpurun_out/<tag>/fetch_probe.json
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-fetch_probe}; mkdir -p "$OUT"; export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/fetch_size_probe.hip -o /tmp/fetch_size_probe
cd /tmp
/tmp/fetch_size_probe > "$OUT/expected.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/f" -o run -- /tmp/fetch_size_probe > /dev/null 2> "$OUT/f.err"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/w" -o run -- /tmp/fetch_size_probe > /dev/null 2> "$OUT/w.err"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys, json
N = 1 << 25
exp = {"k_stream16": 2 * N * 16, "k_stream8": 4 * N * 8, "k_gather32": N * 32 + N * 4, "k_gather32r": N * 32 + N * 4, "k_write8": 0}
expw = {"k_write8": 4 * N * 8}
out = {"_note": "FETCH_SIZE / WRITE_SIZE (KiB as reported, x1024) of tools/fetch_size_probe.hip against the bytes each kernel "
                "really moves once from buffers larger than the Infinity Cache; ratio = reported / real"}
for f in glob.glob(sys.argv[1] + "/*/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    for k, n, cnt, tot in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        name = k.split("(")[0]
        e = (exp if n == "FETCH_SIZE" else expw).get(name)
        out.setdefault(name, {})[n] = {"reported_bytes": tot / cnt * 1024.0, "real_bytes": e, "ratio": (tot / cnt * 1024.0 / e) if e else None}
json.dump(out, open(sys.argv[1] + "/fetch_probe.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find "$OUT" -name '*.db' -size +4M -delete
