#!/usr/bin/env python3
"""Long NVE run on the device loop: energy drift and rebuild bookkeeping over thousands of steps (not a test; run on
the GPU box: python tools/long_run_check.py [unit cells] [steps])."""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imd_amd import lattice, potential  # noqa: E402
from imd_amd.system import GpuSystem, global_cell_dim  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
box = lattice.fcc_box((nb,) * 3)
pos = lattice.fcc_positions((nb,) * 3)
mom = lattice.maxwell_momenta(len(pos), 26.9815, 0.05, seed=3)
pair, rho, emb, r2cut = potential.synthetic_eam_tables()
cellsz = (np.sqrt(r2cut) + 0.5) ** 2
dim = [g + 2 for g in global_cell_dim(box, cellsz)]
s = GpuSystem(box, (1, 1, 1), dim, 1, cellsz, 0.5, 0.1)
s.upload_tables(pair, rho, emb)
s.upload_atoms(pos, mom, 26.9815)
e = []
for st in range(steps):
    s.md_step(st)
    if st % (steps // 10) == 0 or st == steps - 1:
        g = s.get_sums()
        e.append((st, (g["tot_pot_energy"] + g["tot_kin_energy"]) / len(pos), g["tot_kin_energy"] / len(pos), g["nbl_count"]))
        print("step %6d  E/atom %.10f  Ekin/atom %.6f  rebuilds %d" % e[-1], flush=True)
drift = abs(e[-1][1] - e[1][1])
print("atoms %d, drift of E/atom between step %d and %d: %.3e eV (Ekin/atom %.4f)" % (len(pos), e[1][0], e[-1][0], drift, e[-1][2]))
assert drift < 1e-4 * e[-1][2], "energy drift"
s.close()
