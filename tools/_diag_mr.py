# scratch: run the 2x2x2 decomposed-lattice case several times, print the deviations of each run
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_util as gu
G = gu.load("ttm_16384_mpi8")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
extra = dict(e.split("=") for e in sys.argv[2:])
for it in range(n):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res.npz")
        env = dict(os.environ, GPU_MAX_HW_QUEUES="32", PYTHONPATH=ROOT, **extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "local_ranks_worker.py"), "ttm_16384_mpi8", "2", "2", "2", "3", "1", "1", out],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        if r.returncode:
            print("run", it, "rc", r.returncode, r.stdout[-1500:]); continue
        R = dict(np.load(out))
    ids = np.concatenate([R["r%d_nummer" % q] for q in range(8)]); pos = np.concatenate([R["r%d_ort" % q] for q in range(8)])
    o, go = np.argsort(ids), np.argsort(G["s3_nummer"])
    box = np.diag(G["box"])
    dx = pos[o] - G["s3_ort"][go]; dx -= np.round(dx / box) * box
    owner = np.concatenate([np.full(len(R["r%d_nummer" % q]), q) for q in range(8)])[o]
    bad = np.abs(dx).max(axis=1) > 1e-9
    ref = G["t2_substeps_done_temp"][1:-1, 1:-1, 1:-1]
    full = np.zeros_like(ref)
    for q in range(8):
        blk = R["r%d_lat_temp" % q][1:-1, 1:-1, 1:-1]; c, nn = R["r%d_coord" % q], blk.shape
        full[c[0]*nn[0]:(c[0]+1)*nn[0], c[1]*nn[1]:(c[1]+1)*nn[1], c[2]*nn[2]:(c[2]+1)*nn[2]] = blk
    print("run", it, "max|dx| %.3e" % np.abs(dx).max(), "bad atoms", int(bad.sum()), "by rank", np.bincount(owner[bad], minlength=8).tolist(),
          "lat dev %.3e" % (np.abs(full - ref).max() / np.abs(ref).max()),
          "persist", [int(R["r%d_persist" % q]) for q in range(8)], "runs", [int(R["r%d_persist_runs" % q]) for q in range(8)],
          "redone", [int(R["r%d_persist_redone" % q]) for q in range(8)], flush=True)
