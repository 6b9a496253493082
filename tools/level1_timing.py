#!/usr/bin/env python3
"""Throughput of the level-1 drop-in (reference main loop + integrator on the host, forces on the device through the
shim, positions and forces crossing PCIe every step) beside the CPU binary: python tools/level1_timing.py [unit cells]"""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imd_amd import potential  # noqa: E402
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PARAM = """coordname _fcc
outfiles out
ensemble nve
maxsteps 20
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
""" % (nb, nb, nb)
for exe in ("imd_eam_nve_nbl", "imd_eam_nve_nbl_gpu"):
    with tempfile.TemporaryDirectory() as d:
        potential.write_synthetic_eam_files(d)
        open(os.path.join(d, "t.param"), "w").write(PARAM)
        env = dict(os.environ, IMD_PROBE_STEPS="-1", IMD_SHIM_NO_NBL_EXPORT="1")
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        for line in r.stdout.splitlines():
            if "cpuseconds per step and atom" in line or "seconds excluding setup" in line:
                print(exe, line.strip())
