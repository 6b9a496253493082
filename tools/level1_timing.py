#!/usr/bin/env python3
"""Throughput of the level-1 drop-in (reference main loop + integrator on the host, forces on the device through the
shim, positions and forces crossing PCIe every step) beside the CPU binary: python tools/level1_timing.py [unit cells]

WALL time per MD step = (wall of a 45-step run - wall of a 15-step run) / 30: set-up cancels.  (The figure the reference
prints is CPU time summed over threads -- the shim's OpenMP copy loops make that larger while the run gets faster.)"""
import os, sys, subprocess, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imd_amd import potential  # noqa: E402
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PARAM = """coordname _fcc
outfiles out
ensemble nve
maxsteps %d
timestep 0.1
ntypes 1
total_types 1
masses 26.9815
types 0
box_param %d %d %d
box_unit 4.05
pbc_dirs 1 1 1
starttemp 0.0258
seed 4711
core_potential_file phi.pt
embedding_energy_file F.pt
atomic_e-density_file rho.pt
eng_int 100000
checkpt_int 100000
nbl_margin 0.5
nbl_size 1.5
cpu_dim 1 1 1
"""
res = {}
for exe in ("imd_eam_nve_nbl", "imd_eam_nve_nbl_gpu"):
    wall = {}
    for steps in (15, 45):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            open(os.path.join(d, "t.param"), "w").write(PARAM % (steps, nb, nb, nb))
            env = dict(os.environ, IMD_PROBE_STEPS="-1", IMD_SHIM_NO_NBL_EXPORT="1", OMP_WAIT_POLICY="passive")
            t0 = time.time()
            r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            wall[steps] = time.time() - t0
            assert r.returncode == 0, r.stdout[-2000:]
    res[exe] = (wall[45] - wall[15]) / 30.0
    print("%-22s %8.2f ms wall per MD step (%d atoms)" % (exe, 1e3 * res[exe], 4 * nb ** 3))
print("level-1 drop-in / CPU binary on one core: x%.1f" % (res["imd_eam_nve_nbl"] / res["imd_eam_nve_nbl_gpu"]))
