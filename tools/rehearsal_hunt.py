#!/usr/bin/env python3
"""Runs the 2 x 2 x 2 rehearsal (eight ranks as threads on ONE GPU, decomposed TTM lattice, three MD steps:
tests/local_ranks_worker.py ttm_16384_mpi8) again and again until a run's positions differ from the first run's, and keeps both
results (gpurun_out/rehearsal/good.npz, bad.npz, bad.log) with what every step left behind (IMD_TEST_TRACE: momenta, forces, rho,
dF per rank and step).  docs/VARIANTS.md (round 5) has what this found.  Meant to run beside another process that loads the GPU:

    python bench.py --no-cpu-baseline --steps 250000 &      # the load
    python tools/rehearsal_hunt.py 300 [VERIFY=1] [NAME=value ...]

VERIFY=1 sets IMDGPU_LOCAL_VERIFY (every gather, message and unpack of the in-process transport checked on the host; reports on
stderr end up in bad.log and in this script's output); other NAME=value pairs go into the workers' environment."""
import hashlib
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
extra = dict(a.split("=", 1) for a in sys.argv[2:])
if extra.pop("VERIFY", ""):
    extra["IMDGPU_LOCAL_VERIFY"] = "1"
outdir = os.path.join(ROOT, "gpurun_out", "rehearsal")
os.makedirs(outdir, exist_ok=True)
first, t0 = None, time.time()
for it in range(n):
    out = os.path.join(outdir, "cur.npz")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", PYTHONPATH=ROOT, IMD_TEST_TRACE="1", **extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "local_ranks_worker.py"), "ttm_16384_mpi8", "2", "2", "2", "3", "1", "1", out],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    if r.returncode:
        print("run", it, "rc", r.returncode, r.stdout[-1500:], flush=True)
        continue
    if "local verify" in r.stdout:
        print("run", it, "VERIFY:", "\n".join(l for l in r.stdout.split("\n") if "local verify" in l)[:3000], flush=True)
    R = dict(np.load(out))
    order = np.argsort(np.concatenate([R["r%d_nummer" % q] for q in range(8)]))
    pos = np.concatenate([R["r%d_ort" % q] for q in range(8)])[order]
    h = hashlib.md5(np.round(pos, 10).tobytes()).hexdigest()[:8]
    if first is None:
        first = h
        shutil.copy(out, os.path.join(outdir, "good.npz"))
    elif h != first:
        shutil.copy(out, os.path.join(outdir, "bad.npz"))
        with open(os.path.join(outdir, "bad.log"), "w") as f:
            f.write(r.stdout[-20000:])
        G, B = dict(np.load(os.path.join(outdir, "good.npz"))), R
        print("run", it, "DIFFERS", h, "vs", first, "after %.0f s" % (time.time() - t0), flush=True)
        for st in range(3):             # momenta after every step, per rank: largest difference / atoms above 1e-9
            row = []
            for q in range(8):
                d = np.abs(G["r%d_s%d_impuls" % (q, st)] - B["r%d_s%d_impuls" % (q, st)]).max(axis=1)
                row.append("%.1e/%d" % (d.max(), int((d > 1e-9).sum())))
            print("   step", st, " ".join(row), flush=True)
        break
    if it % 20 == 19:
        print("run", it, "same so far, %.0f s" % (time.time() - t0), flush=True)
if os.path.exists(os.path.join(outdir, "cur.npz")):
    os.remove(os.path.join(outdir, "cur.npz"))
