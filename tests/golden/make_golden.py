#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REAL reference.

Runs only in the dev container: needs oracle/_ref/imd_* (built by
oracle/build_ref.sh from /root/reference).  Each case runs the reference
binary on a small synthetic input and stores what oracle/ref_probe.c dumped
from the reference's own memory (full fp64): cell arrangement, positions,
momenta, forces, per-atom energies, rho, dF, Verlet list, global sums and the
potential tables as the reference holds them.

The fixtures are data (inputs + expected outputs); no reference source text.

usage: python tests/golden/make_golden.py [case ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from imd_amd import potential  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

REFBIN = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.dirname(os.path.abspath(__file__))

COMMON = """coordname _fcc
outfiles out
ensemble nve
timestep 0.1
ntypes 1
total_types 1
masses 26.9815
types 0
box_unit 4.05
pbc_dirs {pbc}
seed {seed}
eng_int 1
checkpt_int 100000
cpu_dim {cpu}
"""

CASES = {
    # EAM, Verlet list, serial: config-2 physics at fixture size
    "eam_nbl_2048": dict(binary="imd_eam_nve_nbl", box="8 8 8", temp=0.06, steps=40, dump=[0, 1, 2, 20, 40],
                         full=[0, 40], pot="eam", nbl=True),
    # slab: open boundary in x (pbc_dirs 0 1 1) with vacuum on both sides
    "eam_nbl_slab": dict(binary="imd_eam_nve_nbl", box="8 6 6", temp=0.03, steps=10, dump=[0, 10], full=[0],
                         pot="eam", nbl=True, pbc="0 1 1", extra="shiftx_front 6.0\nshiftx_rear 6.0\n"),
    # LJ, Verlet list, serial: config-1 physics at fixture size
    "lj_nbl_2048": dict(binary="imd_nve_nbl", box="8 8 8", temp=0.0258, steps=20, dump=[0, 20], full=[0],
                        pot="lj", nbl=True),
    # EAM through do_forces/do_forces_eam2 (cell-pair path, no Verlet list)
    "eam_cellpair_2048": dict(binary="imd_eam_nve", box="8 8 8", temp=0.0258, steps=10, dump=[0, 10], full=[0, 10],
                              pot="eam", nbl=False),
}


def run_case(name, c):
    with tempfile.TemporaryDirectory() as d:
        par = COMMON.format(pbc=c.get("pbc", "1 1 1"), seed=c.get("seed", 12345), cpu=c.get("cpu", "1 1 1"))
        par += "box_param %s\nstarttemp %g\nmaxsteps %d\n" % (c["box"], c["temp"], c["steps"])
        if c["pot"] == "eam":
            potential.write_synthetic_eam_files(d)
            par += "core_potential_file phi.pt\nembedding_energy_file F.pt\natomic_e-density_file rho.pt\n"
        else:
            par += "lj_epsilon 0.392\nlj_sigma 2.62\nr_cut 6.0\nr_begin 1.5\npot_res 2000\n"
        if c["nbl"]:
            par += "nbl_margin 0.5\nnbl_size 1.5\n"
        par += c.get("extra", "")
        open(os.path.join(d, "t.param"), "w").write(par)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, c["dump"])))
        r = subprocess.run([os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:])
            raise SystemExit("reference run failed for %s" % name)
        probes = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in c["dump"]}
    P0 = probes[c["dump"][0]]
    out = dict(case=name, param_file=par, steps=np.array(c["dump"]), full_steps=np.array(c["full"]))
    for k in ("ntypes", "cell_dim", "global_cell_dim", "pbc_dirs", "box", "cellsz", "nbl_margin", "timestep",
              "natoms", "flags"):
        out[k] = P0[k]
    for tk in ("pair_pot", "rho_h_tab", "embed_pot"):
        if P0[tk] is not None:
            for f, v in P0[tk].items():
                out["%s_%s" % (tk, f)] = v
    out["cells"] = P0["cells"].astype(np.int32)
    if c["nbl"]:
        out["cnbrs"] = P0["cnbrs"].astype(np.int32)
    for s, P in probes.items():
        pre = "s%d_" % s
        start = np.concatenate([[0], np.cumsum(P["cell_n"])])
        inner_idx = np.concatenate([np.arange(start[cc], start[cc + 1]) for cc in P["cells"]])
        out[pre + "cell_n"] = P["cell_n"].astype(np.int16)
        for k in ("tot_pot_energy", "virial", "tot_kin_energy", "have_valid_nbl", "nbl_count", "last_nbl_len"):
            out[pre + k] = P[k]
        if s in c["full"]:
            out[pre + "all_nummer"] = P["all_nummer"].astype(np.int32)
            out[pre + "all_sorte"] = P["all_sorte"].astype(np.int8)
            out[pre + "all_ort"] = P["all_ort"]
            if c["nbl"]:
                out[pre + "tl"] = P["tl"]; out[pre + "tb"] = P["tb"]; out[pre + "cl_off"] = P["cl_off"]
        out[pre + "nummer"] = P["all_nummer"][inner_idx].astype(np.int32)
        out[pre + "sorte"] = P["all_sorte"][inner_idx].astype(np.int8)
        out[pre + "ort"] = P["all_ort"][inner_idx]
        for k in ("masse", "impuls", "kraft", "pot_eng", "eam_rho", "eam_dF", "nbl_pos", "numneighs"):
            if k in ("eam_rho", "eam_dF") and c["pot"] != "eam":
                continue
            if k == "nbl_pos" and not c["nbl"]:
                continue
            out[pre + k] = P[k]
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB  steps=%s nbl_count(last)=%d maxF=%.3g" %
          (name, os.path.getsize(path) / 1024, c["dump"], probes[c["dump"][-1]]["nbl_count"],
           np.abs(probes[c["dump"][-1]]["kraft"]).max()))


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n, CASES[n])
