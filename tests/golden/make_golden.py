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
import shutil  # noqa: E402
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
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
    # one atom type with phi, rho and F on THREE DIFFERENT grids (begin, end, step, row count; potential.EAM_GRIDS["distinct"]):
    # separate files with separate headers, the normal state of hand-made EAM tables (read_pot_table2, imd_potential.c:394-466)
    "eam_nbl_2048_dg": dict(binary="imd_eam_nve_nbl", box="8 8 8", temp=0.06, steps=40, dump=[0, 1, 2, 20, 40],
                            full=[0, 40], pot="eam_dg", nbl=True),
    # ... and with the 3D TTM (neighbour counts against pair_pot.end, the embedding pass folded into pass 1, both EAM force kernels)
    "ttm_2048_dg": dict(binary="imd_mpi_eam_ttm3d_nbl", box="8 8 8", temp=0.03, steps=3, dump=[0, 1, 2, 3], full=[],
                        pot="eam_dg", nbl=True, ttm=dict(fd_ext="1 1 1", n=100, min_atoms=5)),
    # slab: open boundary in x (pbc_dirs 0 1 1) with vacuum on both sides
    "eam_nbl_slab": dict(binary="imd_eam_nve_nbl", box="8 6 6", temp=0.03, steps=10, dump=[0, 10], full=[0],
                         pot="eam", nbl=True, pbc="0 1 1", extra="shiftx_front 6.0\nshiftx_rear 6.0\n"),
    # LJ, Verlet list, serial: config-1 physics at fixture size
    "lj_nbl_2048": dict(binary="imd_nve_nbl", box="8 8 8", temp=0.0258, steps=20, dump=[0, 20], full=[0],
                        pot="lj", nbl=True),
    # pair potential from a FORMAT 1 file (read_pot_table1, imd_potential.c:297-375)
    "pair_fmt1": dict(binary="imd_nve_nbl", box="4 4 4", temp=0.0258, steps=0, dump=[0], full=[], pot="pair1", nbl=True),
    # EAM through do_forces/do_forces_eam2 (cell-pair path, no Verlet list)
    # EAM + 3D TTM (imd_ttm_3D.c functions driven by oracle/ref_probe_ttm.c), laser off
    "ttm_2048": dict(binary="imd_mpi_eam_ttm3d_nbl", box="8 8 8", temp=0.03, steps=3, dump=[0, 1, 2, 3], full=[],
                     pot="eam", nbl=True, ttm=dict(fd_ext="1 1 1", n=100, min_atoms=5)),
    "ttm_6912_ext2": dict(binary="imd_mpi_eam_ttm3d_nbl", box="12 12 12", temp=0.0258, steps=1, dump=[0, 1], full=[],
                          pot="eam", nbl=True, ttm=dict(fd_ext="2 2 2", n=25, min_atoms=5)),
    # slab with vacuum in x: inactive FD cells, mirrored faces, open boundary
    "ttm_slab": dict(binary="imd_mpi_eam_ttm3d_nbl", box="8 6 6", temp=0.03, steps=2, dump=[0, 1, 2], full=[],
                     pot="eam", nbl=True, pbc="0 1 1", extra="shiftx_front 8.0\nshiftx_rear 8.0\n",
                     ttm=dict(fd_ext="1 1 1", n=100, min_atoms=12)),
    # Beer-Lambert laser source (laser_rescale_ttm, imd_laser.c:715-788, -DLASER build) into the slab: pulse centred on
    # step 3 (t_0 0.3, sigma_t 0.15 in IMD time units), absorption length 20 A; the source term then drives do_DIFF
    "ttm_laser": dict(binary="imd_mpi_eam_ttm3d_laser_nbl", box="8 6 6", temp=0.03, steps=5, dump=[0, 2, 3, 5], full=[],
                      pot="eam", nbl=True, pbc="0 1 1",
                      extra="shiftx_front 8.0\nshiftx_rear 8.0\nlaser_dir 1 0 0\nlaser_mu 0.05\nlaser_sigma_e 40.0\n"
                            "laser_rescale_mode 4\n",
                      ttm=dict(fd_ext="1 1 1", n=100, min_atoms=12, laser_sigma_t="0.15", laser_t_0="0.3"),
                      env=dict(IMD_PROBE_LASER="1"), keep_stages=("update_fd", "laser", "substep1", "substeps_done")),
    # two atom types on the Cu3Au (L1_2) lattice: 4-column pair / density tables with per-column ranges
    "eam2_l12_nbl": dict(binary="imd_eam_nve_nbl", box="8 8 8", temp=0.05, steps=20, dump=[0, 1, 20], full=[0],
                         pot="eam2types", nbl=True, coord="_l12",
                         types="ntypes 2\ntotal_types 2\nmasses 26.9815 63.546\ntypes 0 1"),
    # the same with coarse tables (600 rows): they fit the LDS, so the TILE kernels' several-type branch runs
    "eam2_l12_coarse": dict(binary="imd_eam_nve_nbl", box="8 8 8", temp=0.05, steps=20, dump=[0, 20], full=[0],
                            pot="eam2types_coarse", nbl=True, coord="_l12",
                            types="ntypes 2\ntotal_types 2\nmasses 26.9815 63.546\ntypes 0 1"),
    # two atom types WITH the TTM: update_fd's density counts only pairs of type 0 (imd_forces_nbl.c:473-476)
    "ttm_2types": dict(binary="imd_mpi_eam_ttm3d_nbl", box="8 8 8", temp=0.04, steps=2, dump=[0, 1, 2], full=[],
                       pot="eam2types_coarse", nbl=True, coord="_l12",
                       types="ntypes 2\ntotal_types 2\nmasses 26.9815 63.546\ntypes 0 1",
                       ttm=dict(fd_ext="1 1 1", n=100, min_atoms=5)),
    # atom-flux counters of update_fd (imd_ttm_3D.c:335-370): 19 x 8 x 6 unit cells put lattice planes 0.2 A from
    # cell boundaries, so atoms change FD cells at the list rebuilds of steps 17, 30 and 43; one FD cell thick in z
    "ttm_flux": dict(binary="imd_mpi_eam_ttm3d_nbl", box="19 8 6", temp=0.12, steps=44, dump=[0, 1, 17, 30, 43], full=[],
                     pot="eam", nbl=True, ttm=dict(fd_ext="1 1 3", n=100, min_atoms=5), env=dict(IMD_PROBE_FLUX="1")),
    # do_ADV / do_COMMFLUX (imd_ttm_3D.c:1321-1458, :667-995): the same system on TWO ranks along x -- the reference's
    # advection step only runs with cpu_dim.x >= 2 (on one rank do_COMMFLUX posts a self-Sendrecv whose tags never
    # match) -- assembled into the global lattice; lattice stages only (see run_adv_case)
    "ttm_adv": dict(binary="imd_mpi_eam_ttm3d_nbl", box="19 8 6", temp=0.12, steps=60, dump=None, full=[],
                    pot="eam", nbl=True, cpu="2 1 1", ttm=dict(fd_ext="1 1 3", n=100, min_atoms=5),
                    env=dict(IMD_PROBE_ADV="1")),
    # BASELINE config 4's layout at fixture size: the reference itself on 2x2x2 ranks (mpiexec -n 8), 16 384 atoms, every
    # rank's memory dumped and assembled into global arrays (SURVEY 8c golden (1)); see run_mpi_case
    "eam_nbl_16384_mpi8": dict(binary="imd_mpi_eam_nve_nbl", box="16 16 16", temp=0.06, steps=20, dump=[0, 20], full=[],
                               pot="eam", nbl=True, cpu="2 2 2", extra="msgbuf_size 3.0\n"),
    # the same grid with the 3D TTM: eight lattice blocks of 4^3 FD cells with MPI ghost faces, assembled into the 8^3 lattice
    "ttm_16384_mpi8": dict(binary="imd_mpi_eam_ttm3d_nbl", box="16 16 16", temp=0.03, steps=3, dump=[0, 1, 2, 3], full=[],
                           pot="eam", nbl=True, cpu="2 2 2", extra="msgbuf_size 3.0\n", ttm=dict(fd_ext="1 1 1", n=100, min_atoms=5)),
    # ... and with do_ADV / do_COMMFLUX on that grid (the headline's sequence on config 4's geometry: a slab with open x on
    # 2 x 2 x 2 ranks).  24 unit cells + 4.5 / 3.5 half lattice constants of vacuum in front / behind: 16 MD = FD cells of 3.5 half
    # lattice constants along x, so every second cell boundary -- the boundary between the two ranks along x among them -- lies ON
    # a lattice plane and atoms change FD cells, and ranks, in both directions at every list rebuild; 8 x 4 x 4 FD cells per rank,
    # surface cells switch on and off; every rank advects the plane k = 1 of ITS block (imd_ttm_3D.c:1327), do_COMMFLUX
    # carries {flux, U} across the rank boundary (:667-995).  All steps are dumped, a few kept.
    "ttm_adv_mpi8": dict(binary="imd_mpi_eam_ttm3d_nbl", box="24 16 16", temp=0.12, steps=27, dump=None, full=[],
                         pot="eam", nbl=True, cpu="2 2 2", pbc="0 1 1", adv=True,
                         extra="msgbuf_size 3.0\nshiftx_front 9.1125\nshiftx_rear 7.0875\n",
                         ttm=dict(fd_ext="1 1 1", n=100, min_atoms=5), env=dict(IMD_PROBE_ADV="1")),
    # a slab with vacuum on TWO ranks along x, 3D TTM: 10 x 4 x 4 FD cells, FIVE layers per rank (an extent of 4 q + 1) of which
    # the outer ones are empty -- what a rank of BASELINE config 4 holds (45 x 39 x 39 FD cells, six layers of vacuum)
    "ttm_slab_mpi2": dict(binary="imd_mpi_eam_ttm3d_nbl", box="14 8 8", temp=0.03, steps=3, dump=[0, 1, 2, 3], full=[],
                          pot="eam", nbl=True, cpu="2 1 1", pbc="0 1 1", extra="msgbuf_size 3.0\nshiftx_front 8.0\nshiftx_rear 8.0\n",
                          ttm=dict(fd_ext="1 1 1", n=100, min_atoms=5)),
    "eam_cellpair_2048": dict(binary="imd_eam_nve", box="8 8 8", temp=0.0258, steps=10, dump=[0, 10], full=[0, 10],
                              pot="eam", nbl=False),
}


def write_planar_eos(d):
    """Synthetic EOS tables, planar in the table's y argument so that linear
    interpolation on ANY triangulation (the reference's nn-c lpi) reproduces them:
    cve(r,T) = 0.05*T J/(kg K); ee(r,sqrt T) = 4000*sqrt(T) J/kg."""
    nr, nt = 24, 40
    r = np.linspace(500.0, 4500.0, nr); t = np.linspace(1.0, 60000.0, nt); ts = np.linspace(1.0, 245.0, nt)
    zc = np.outer(np.ones(nr), 0.05 * t); ze = np.outer(np.ones(nr), 4000.0 * ts)
    for fn, y, z in (("EOS_cve_from_r_te.txt", t, zc), ("EOS_ee_from_r_tesqrt.txt", ts, ze)):
        with open(os.path.join(d, fn), "w") as f:
            f.write("%d %d\n%.16e %.16e %.16e %.16e\n" % (nr, nt, r[0], r[-1], y[0], y[-1]))
            for i in range(nr):
                for j in range(nt):
                    f.write("%.16e %.16e %.16e\n" % (r[i], y[j], z[i, j]))
    return dict(nr=nr, nt=nt, r=r, t=t, ts=ts, cve=zc, ee=ze)


def write_inputs(c, d):
    """parameter file, potential files and EOS tables of case `c` in directory `d`"""
    eos = None
    par = COMMON.format(pbc=c.get("pbc", "1 1 1"), seed=c.get("seed", 12345), cpu=c.get("cpu", "1 1 1"))
    par += "box_param %s\nstarttemp %g\nmaxsteps %d\n" % (c["box"], c["temp"], c["steps"])
    if "coord" in c:
        par = par.replace("coordname _fcc", "coordname " + c["coord"])
    if "types" in c:
        par = par.replace("ntypes 1\ntotal_types 1\nmasses 26.9815\ntypes 0", c["types"])
    if c["pot"].startswith("eam2types"):
        potential.write_synthetic_eam_files_2types(d, rows=0.15 if c["pot"].endswith("coarse") else 1.0)
        par += "core_potential_file phi.pt\nembedding_energy_file F.pt\natomic_e-density_file rho.pt\n"
    elif c["pot"] in ("eam", "eam_dg"):
        potential.write_synthetic_eam_files(d, grids="distinct" if c["pot"] == "eam_dg" else "shared")
        par += "core_potential_file phi.pt\nembedding_energy_file F.pt\natomic_e-density_file rho.pt\n"
    elif c["pot"] == "pair1":       # the LJ table of config 1, handed over as a format-1 potential file
        pt, _ = potential.create_lj_table(0.392, 2.62, 6.0, 1.5, 731)
        r2 = pt.begin[0] + pt.step[0] * np.arange(731)
        potential.write_pot_table1(os.path.join(d, "pair1.pt"), r2, [pt.table[:731]])
        par += "potfile pair1.pt\n"
    else:
        par += "lj_epsilon 0.392\nlj_sigma 2.62\nr_cut 6.0\nr_begin 1.5\npot_res 2000\n"
    if c["nbl"]:
        par += "nbl_margin 0.5\nnbl_size 1.5\n"
    par += c.get("extra", "")
    if "ttm" in c:
        par = par.replace("ensemble nve", "ensemble ttm")
        par += ("fd_ext %s\nfd_n_timesteps %d\nfd_min_atoms %d\natomic_weight 26.9815\natomic_charge 13\n"
                "ttm_int 100000\nI0 1.0\nlambda 800e-9\nlaser_sigma_t %s\nlaser_t_0 %s\n"
                % (c["ttm"]["fd_ext"], c["ttm"]["n"], c["ttm"]["min_atoms"], c["ttm"].get("laser_sigma_t", "100e-15"),
                   c["ttm"].get("laser_t_0", "1.0")))
        eos = write_planar_eos(d)
    open(os.path.join(d, "t.param"), "w").write(par)
    return par, eos


def run_case(name, c):
    with tempfile.TemporaryDirectory() as d:
        par, eos = write_inputs(c, d)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, c["dump"])))
        env.update(c.get("env", {}))
        r = subprocess.run([os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:])
            raise SystemExit("reference run failed for %s" % name)
        probes = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in c["dump"]}
        tprobes = {s: po.read_ttm_probe(os.path.join(d, "ttm.0.%d.bin" % s)) for s in c["dump"]} if "ttm" in c else {}
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
            if k in ("eam_rho", "eam_dF") and not c["pot"].startswith("eam"):
                continue
            if k == "nbl_pos" and not c["nbl"]:
                continue
            out[pre + k] = P[k]
    for s, T in tprobes.items():
        if s == c["dump"][0]:
            for k in ("fd_dim", "fd_ext", "fd_min_atoms", "diff_substeps", "fd_n_timesteps", "fd_h", "fd_vol", "neighvol",
                      "atomic_weight", "atomic_charge", "tau", "omega_laser"):
                out["ttm_" + k] = T[k]
            for k, v in eos.items():
                out["eos_" + k] = v
        for tag, st in T["stages"].items():
            if "flux" in st:
                out["t%d_flux" % s] = st["flux"].astype(np.int16)
                continue
            if "eabs" in st:
                out["t%d_eabs" % s] = st["eabs"]; out["t%d_laser_active" % s] = st["laser_active"]
                continue
            if "params" in st:
                for k, v in st["params"].items():
                    out["laser_" + k] = v
                continue
            if "env" in c and tag not in c.get("keep_stages", ("update_fd", "substeps_done")):
                continue                                  # fixtures of the optional stages keep few stages per step
            out["t%d_%s_max_dt_ttm" % (s, tag)] = st["max_dt_ttm"]
            for f in po.TTM_REC.names:
                if tag in ("entry", "substep2") and f not in ("temp", "xi", "natoms", "fd_k", "U", "temp2"):
                    continue
                out["t%d_%s_%s" % (s, tag, f)] = st["lat"][f]
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB  steps=%s nbl_count(last)=%d maxF=%.3g" %
          (name, os.path.getsize(path) / 1024, c["dump"], probes[c["dump"][-1]]["nbl_count"],
           np.abs(probes[c["dump"][-1]]["kraft"]).max()))


def run_eos_inverse():
    """EOS_te_from_r_ee / EOS_ee_from_r_te of the reference (Brent minimiser over the ee(r, sqrt T) table) on the
    sample points of oracle/ref_probe_ttm.c:dump_eos_inverse -> tests/golden/eos_inverse.npz"""
    c = CASES["ttm_2048"]
    with tempfile.TemporaryDirectory() as d:
        par = COMMON.format(pbc="1 1 1", seed=12345, cpu="1 1 1")
        par += "box_param %s\nstarttemp %g\nmaxsteps 0\n" % (c["box"], c["temp"])
        potential.write_synthetic_eam_files(d)
        par += "core_potential_file phi.pt\nembedding_energy_file F.pt\natomic_e-density_file rho.pt\n"
        par += "nbl_margin 0.5\nnbl_size 1.5\n"
        par = par.replace("ensemble nve", "ensemble ttm")
        par += ("fd_ext 1 1 1\nfd_n_timesteps 100\nfd_min_atoms 5\natomic_weight 26.9815\natomic_charge 13\n"
                "ttm_int 100000\nI0 1.0\nlambda 800e-9\nlaser_sigma_t 100e-15\nlaser_t_0 1.0\n")
        eos = write_planar_eos(d)
        open(os.path.join(d, "t.param"), "w").write(par)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0", IMD_PROBE_EOSINV="1")
        r = subprocess.run([os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:])
            raise SystemExit("reference run failed for eos_inverse")
        raw = open(os.path.join(d, "eosinv.0.bin"), "rb").read()
    assert raw[:8] == b"IMDEOSIN"
    n = int(np.frombuffer(raw[8:12], np.int32)[0])
    a = np.frombuffer(raw[12:12 + 32 * n], np.float64).reshape(n, 4)
    out = dict(r=a[:, 0], t_in=a[:, 1], e=a[:, 2], t_out=a[:, 3])
    for k, v in eos.items():
        out["eos_" + k] = v
    path = os.path.join(OUT, "eos_inverse.npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB  %d points, max |t_out/t_in - 1| on the curve = %.3g" %
          ("eos_inverse", os.path.getsize(path) / 1024, n, np.abs(a[::3, 3] / a[::3, 1] - 1).max()))


def _curved_tables(nr, nt):
    """Curved, NON-separable synthetic tables on a regular (r, y) grid; both tables share the grid so that one sample
    file serves both interpolators of the probe.  Shapes loosely follow the physics (C_e/rho rising then saturating in
    T with a density-dependent knee; e growing like T^2 -> T) -- only the curvature matters here."""
    r = np.linspace(500.0, 4500.0, nr)
    y = np.linspace(1.0, 245.0, nt)
    R, Y = np.meshgrid(r, y, indexing="ij")
    zc = 40.0 * Y * (1.0 + 0.3 * np.sin(R / 700.0)) / (1.0 + (Y / (60.0 + R / 50.0)) ** 2) + 1e-3 * R
    ze = 900.0 * Y * Y / (1.0 + Y / (30.0 + 0.02 * R)) * (1.0 + 0.2 * np.cos(R / 900.0 + Y / 80.0)) + 5.0 * R
    return r, y, zc, ze


def run_eos_grid():
    """SURVEY 8(c) / VERDICT r1 item 2: the reference's own table interpolators (nn-c lpi_interpolate_point behind
    EOS_cve_from_r_te / EOS_ee_from_r_te, imd_ttm_3D.c:2722,2759) evaluated on CURVED tables at every grid node, every
    cell-edge midpoint, every cell centre and two random interior points per cell (probe option IMD_PROBE_EOSGRID,
    oracle/ref_probe_ttm.c:dump_eos_grid) -> tests/golden/eos_grid.npz.  The MD run itself is a throw-away 2048-atom
    step 0; only the interpolator dump is kept."""
    c = CASES["ttm_2048"]
    out = {}
    for tag, (nr, nt) in (("a", (9, 11)), ("b", (24, 40))):
        r, y, zc, ze = _curved_tables(nr, nt)
        rng = np.random.default_rng(100 + nr)
        pts, kind = [], []
        for i in range(nr):
            for j in range(nt):
                pts.append((r[i], y[j])); kind.append(0)                                   # node
                if i + 1 < nr:
                    pts.append((0.5 * (r[i] + r[i + 1]), y[j])); kind.append(1)            # edge along r
                if j + 1 < nt:
                    pts.append((r[i], 0.5 * (y[j] + y[j + 1]))); kind.append(1)            # edge along y
                if i + 1 < nr and j + 1 < nt:
                    pts.append((0.5 * (r[i] + r[i + 1]), 0.5 * (y[j] + y[j + 1]))); kind.append(2)   # cell centre
                    for _ in range(2):
                        u, v = rng.uniform(0.02, 0.98, 2)
                        pts.append((r[i] + u * (r[i + 1] - r[i]), y[j] + v * (y[j + 1] - y[j]))); kind.append(3)
        pts = np.array(pts); kind = np.array(kind, np.int8)
        with tempfile.TemporaryDirectory() as d:
            par = COMMON.format(pbc="1 1 1", seed=12345, cpu="1 1 1")
            par += "box_param %s\nstarttemp %g\nmaxsteps 0\n" % (c["box"], c["temp"])
            potential.write_synthetic_eam_files(d)
            par += "core_potential_file phi.pt\nembedding_energy_file F.pt\natomic_e-density_file rho.pt\n"
            par += "nbl_margin 0.5\nnbl_size 1.5\n"
            par = par.replace("ensemble nve", "ensemble ttm")
            par += ("fd_ext 1 1 1\nfd_n_timesteps 1\nfd_min_atoms 5\natomic_weight 26.9815\natomic_charge 13\n"
                    "ttm_int 100000\nI0 1.0\nlambda 800e-9\nlaser_sigma_t 100e-15\nlaser_t_0 1.0\n")
            for fn, z in (("EOS_cve_from_r_te.txt", zc), ("EOS_ee_from_r_tesqrt.txt", ze)):
                with open(os.path.join(d, fn), "w") as f:
                    f.write("%d %d\n%.16e %.16e %.16e %.16e\n" % (nr, nt, r[0], r[-1], y[0], y[-1]))
                    for i in range(nr):
                        for j in range(nt):
                            f.write("%.16e %.16e %.16e\n" % (r[i], y[j], z[i, j]))
            with open(os.path.join(d, "eos_samples.txt"), "w") as f:
                for x, yy in pts:
                    f.write("%.17g %.17g\n" % (x, yy))
            open(os.path.join(d, "t.param"), "w").write(par)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0", IMD_PROBE_EOSGRID="1")
            rr = subprocess.run([os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            fn = os.path.join(d, "eosgrid.0.bin")
            if not os.path.exists(fn):
                print(rr.stdout[-3000:])
                raise SystemExit("reference run failed for eos_grid")
            a = np.fromfile(fn, np.float64).reshape(-1, 4)
        assert len(a) == len(pts) and np.array_equal(a[:, :2], pts)
        out.update({tag + "_r": r, tag + "_y": y, tag + "_cve": zc, tag + "_ee": ze, tag + "_x": a[:, 0],
                    tag + "_yq": a[:, 1], tag + "_kind": kind, tag + "_ref_cve": a[:, 2], tag + "_ref_ee": a[:, 3]})
        print("eos_grid %s: %dx%d table, %d samples, %d NaN from the reference" %
              (tag, nr, nt, len(a), int(np.isnan(a[:, 2:]).any(axis=1).sum())))
    path = os.path.join(OUT, "eos_grid.npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB" % ("eos_grid", os.path.getsize(path) / 1024))


FDTD_ARGS = dict(nx=160, hx=7.0, lo=60, hi=140, I0=1e16, lam=800e-9, sigma_t=1e-15, t0=3e-15, srcx=2.1e-8, bw=10, Sc=1.0,
                 nmd=6, timestep=0.1)


def run_fdtd_case():
    """SURVEY 8(f)2 / BASELINE config 5: the reference's own 1D Maxwell solver (do_fdtd, fdtd_softsource, init_fdtd,
    init_pml; imd_fdtd.c) driven at function level by oracle/ref_probe_fdtd.c (its caller imd_ttm_3D.c does not compile
    with -DFDTD as shipped) -> tests/golden/fdtd_1d.npz: a 158-cell row, 7 A cells, an 80-cell Al slab behind 60 cells
    of vacuum, a 1 fs pulse of 1e12 W/cm^2 at 800 nm, six MD steps of 437 FDTD sub-steps each."""
    a = FDTD_ARGS
    exe = os.path.join(REFBIN, "ref_probe_fdtd")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "fdtd.bin")
        r = subprocess.run([exe, out] + [repr(a[k]) for k in ("nx", "hx", "lo", "hi", "I0", "lam", "sigma_t", "t0", "srcx", "bw",
                                                               "Sc", "nmd", "timestep")],
                           cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:]); raise SystemExit("ref_probe_fdtd failed")
        H = po.read_fdtd_probe(out)
    o = {"arg_" + k: v for k, v in a.items()}
    for k in ("nx", "bw", "srcx_int", "containssrc", "dt_FDTD", "fd_dx", "E0", "E0_init", "Nlambdax", "Imp0", "sigma_t_squared",
              "omega_laser"):
        o[k] = H[k]
    for k in ("natoms", "dens", "temp", "ne", "DL"):
        o["cell_" + k] = H["cells"][k]
    o["md_nsub"] = np.array([m[0] for m in H["md"]]); o["md_active"] = np.array([m[1] for m in H["md"]])
    for tag, dmp in H["dumps"].items():
        for k, v in dmp.items():
            o["%s_%s" % (tag, k)] = v
    path = os.path.join(OUT, "fdtd_1d.npz")
    np.savez_compressed(path, **o)
    print("%-20s %8.1f KB  dumps: %s" % ("fdtd_1d", os.path.getsize(path) / 1024, " ".join(H["dumps"])))


def run_adv_case(name, c):
    """do_ADV fixture: the reference on TWO ranks along x (mpiexec -n 2), every rank's lattice dumped by
    oracle/ref_probe_ttm.c around update_fd / do_ADV / the sub-steps and assembled here into the GLOBAL lattice
    (rank r's inner columns 1..lx-2 become global columns r*(lx-2)+1.., the outer ghost columns are kept), plus the
    global step-0 atom arrangement in the single-rank layout the other fixtures use."""
    nsteps = c["steps"]
    with tempfile.TemporaryDirectory() as d:
        par, eos = write_inputs(c, d)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, range(nsteps + 1))))
        env.update(c["env"])
        r = subprocess.run([MPIEXEC, "-n", "2", os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:])
            raise SystemExit("reference run failed for %s" % name)
        T = [[po.read_ttm_probe(os.path.join(d, "ttm.%d.%d.bin" % (rk, s))) for rk in (0, 1)] for s in range(nsteps + 1)]
        P = [po.read_probe(os.path.join(d, "probe.%d.0.bin" % rk)) for rk in (0, 1)]

    def glob(a0, a1):                   # two local arrays [lx, ...] -> global [2(lx-2)+2, ...]
        return np.concatenate([a0[:-1], a1[1:]], axis=0)

    fluxsum = [int(T[s][0]["stages"]["flux"]["flux"][1:-1].sum() + T[s][1]["stages"]["flux"]["flux"][1:-1].sum())
               for s in range(nsteps + 1)]
    moved = [s for s in range(nsteps + 1) if fluxsum[s] > 0]
    assert len(moved) >= 2, "no atom changed its FD cell: nothing to advect"
    keep = sorted(set([0, 1, 2, moved[0] - 1, moved[0], moved[0] + 1, moved[1], moved[1] + 1]))
    P0 = P[0]
    out = dict(case=name, param_file=par, steps=np.array(keep), flux_steps=np.array(moved), full_steps=np.array([], int))
    for k in ("ntypes", "pbc_dirs", "box", "cellsz", "nbl_margin", "timestep", "global_cell_dim", "flags"):
        out[k] = P0[k]
    for tk in ("pair_pot", "rho_h_tab", "embed_pot"):
        for f, v in P0[tk].items():
            out["%s_%s" % (tk, f)] = v
    # global step-0 arrangement: inner cells in x-outer order, each cell's atoms in its owner's slot order
    ld = P0["cell_dim"]; nx = ld[0] - 2
    gd = np.array([2 * nx + 2, ld[1], ld[2]])
    out["cell_dim"] = gd
    out["natoms"] = P[0]["ninner_atoms"] + P[1]["ninner_atoms"]
    cell_n = np.zeros(int(np.prod(gd)), np.int64)
    cells, cols = [], {k: [] for k in ("nummer", "sorte", "ort", "masse", "impuls")}
    for gi in range(1, gd[0] - 1):
        rk, li = (gi - 1) // nx, (gi - 1) % nx + 1
        Pr = P[rk]
        start = np.concatenate([[0], np.cumsum(Pr["cell_n"])])
        istart = np.concatenate([[0], np.cumsum(Pr["cell_n"][Pr["cells"]])])
        pos_in_cells = {int(cc): q for q, cc in enumerate(Pr["cells"])}
        for j in range(1, gd[1] - 1):
            for k in range(1, gd[2] - 1):
                lc = (li * ld[1] + j) * ld[2] + k
                gc = (gi * gd[1] + j) * gd[2] + k
                m = int(Pr["cell_n"][lc]); cell_n[gc] = m; cells.append(gc)
                q = pos_in_cells[lc]
                cols["nummer"].append(Pr["all_nummer"][start[lc]:start[lc] + m])
                cols["sorte"].append(Pr["all_sorte"][start[lc]:start[lc] + m])
                cols["ort"].append(Pr["all_ort"][start[lc]:start[lc] + m])
                cols["masse"].append(Pr["masse"][istart[q]:istart[q] + m])
                cols["impuls"].append(Pr["impuls"][istart[q]:istart[q] + m])
    out["cells"] = np.array(cells, np.int32)
    out["s0_cell_n"] = cell_n.astype(np.int16)
    out["s0_nummer"] = np.concatenate(cols["nummer"]).astype(np.int32)
    out["s0_sorte"] = np.concatenate(cols["sorte"]).astype(np.int8)
    out["s0_ort"] = np.concatenate(cols["ort"]); out["s0_masse"] = np.concatenate(cols["masse"])
    out["s0_impuls"] = np.concatenate(cols["impuls"])
    T0 = T[0][0]
    for k in ("fd_ext", "fd_min_atoms", "diff_substeps", "fd_n_timesteps", "fd_h", "fd_vol", "neighvol",
              "atomic_weight", "atomic_charge", "tau", "omega_laser"):
        out["ttm_" + k] = T0[k]
    out["ttm_fd_dim"] = np.array([2 * (T0["fd_dim"][0] - 2) + 2, T0["fd_dim"][1], T0["fd_dim"][2]])
    for k, v in eos.items():
        out["eos_" + k] = v
    for s in keep:
        st0, st1 = T[s][0]["stages"], T[s][1]["stages"]
        out["t%d_flux" % s] = glob(st0["flux"]["flux"], st1["flux"]["flux"]).astype(np.int16)
        for tag in ("update_fd", "adv", "fillmesh0", "substeps_done"):
            for f in po.TTM_REC.names:
                if tag == "fillmesh0" and f not in ("temp", "U", "temp2", "U2", "natoms"):
                    continue
                out["t%d_%s_%s" % (s, tag, f)] = glob(st0[tag]["lat"][f], st1[tag]["lat"][f])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB  kept steps=%s  steps with flux=%s" % (name, os.path.getsize(path) / 1024, keep, moved))


def run_mpi_case(name, c):
    """The reference on a rank grid (mpiexec -n P): every rank's probe dump assembled into GLOBAL arrays.
    Atoms: the inner atoms of all ranks concatenated (rank order, cell order inside a rank) -- positions, momenta, forces,
    energies by atom number; sums are the reference's all-reduced globals.  TTM: every rank's lattice block put at its place
    in the global lattice (inner cells; the outer ghost shell stays zero)."""
    cpu = [int(v) for v in c["cpu"].split()]
    world = cpu[0] * cpu[1] * cpu[2]
    adv = bool(c.get("adv"))
    if adv:
        c = dict(c, dump=list(range(c["steps"] + 1)))
    with tempfile.TemporaryDirectory() as d:
        par, eos = write_inputs(c, d)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, c["dump"])))
        env.update(c.get("env", {}))
        r = subprocess.run([MPIEXEC, "-n", str(world), os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:])
            raise SystemExit("reference run failed for %s" % name)
        P = {s: [po.read_probe(os.path.join(d, "probe.%d.%d.bin" % (rk, s))) for rk in range(world)] for s in c["dump"]}
        T = {s: [po.read_ttm_probe(os.path.join(d, "ttm.%d.%d.bin" % (rk, s))) for rk in range(world)] for s in c["dump"]} if "ttm" in c else {}
    if adv:            # keep the first and the last step, and the steps at which atoms changed FD cells (with the step before each)
        fluxsum = {s: sum(int(T[s][rk]["stages"]["flux"]["flux"][1:-1, 1:-1, 1:-1].sum()) for rk in range(world)) for s in c["dump"]}
        moved = [s for s in c["dump"] if fluxsum[s] > 0]
        assert len(moved) >= 2 and moved[-1] < c["dump"][-1], "atoms must change FD cells at two steps at least: %s" % moved
        keep = sorted(set([0, c["dump"][-1]] + [s for m in moved for s in (m - 1, m, m + 1) if 0 <= s <= c["dump"][-1]]))
        P = {s: P[s] for s in keep}; T = {s: T[s] for s in keep}
        c = dict(c, dump=keep)
    P0 = P[c["dump"][0]][0]
    out = dict(case=name, param_file=par, steps=np.array(c["dump"]), full_steps=np.array([], int), cpu_dim=np.array(cpu))
    for k in ("ntypes", "global_cell_dim", "pbc_dirs", "box", "cellsz", "nbl_margin", "timestep", "natoms", "flags"):
        out[k] = P0[k]
    out["cell_dim"] = np.array(P0["global_cell_dim"]) + 2              # the single-domain layout the other fixtures use
    for tk in ("pair_pot", "rho_h_tab", "embed_pot"):
        if P0[tk] is not None:
            for f, v in P0[tk].items():
                out["%s_%s" % (tk, f)] = v
    for s in c["dump"]:
        pre = "s%d_" % s
        for k in ("tot_pot_energy", "virial", "tot_kin_energy", "nbl_count"):
            out[pre + k] = P[s][0][k]
        if s not in (c["dump"][0], c["dump"][-1]):
            continue                                    # per-atom arrays of the first and the last dumped step only
        cols = {k: [] for k in ("nummer", "sorte", "ort", "masse", "impuls", "kraft", "pot_eng", "eam_rho", "eam_dF", "numneighs", "rank")}
        for rk, Pr in enumerate(P[s]):
            assert tuple(Pr["cpu_dim"]) == tuple(cpu) and Pr["myid"] == rk
            start = np.concatenate([[0], np.cumsum(Pr["cell_n"])])
            inner_idx = np.concatenate([np.arange(start[cc], start[cc + 1]) for cc in Pr["cells"]] + [np.zeros(0, int)]).astype(int)
            cols["nummer"].append(Pr["all_nummer"][inner_idx]); cols["sorte"].append(Pr["all_sorte"][inner_idx])
            cols["ort"].append(Pr["all_ort"][inner_idx])
            for k in ("masse", "impuls", "kraft", "pot_eng", "eam_rho", "eam_dF", "numneighs"):
                cols[k].append(Pr[k])
            cols["rank"].append(np.full(len(inner_idx), rk))
        for k, v in cols.items():
            a = np.concatenate(v)
            out[pre + k] = a.astype(np.int32) if k in ("nummer", "numneighs") else a.astype(np.int8) if k in ("sorte", "rank") else a
        assert len(np.unique(out[pre + "nummer"])) == out["natoms"]
    if T:
        T0 = T[c["dump"][0]][0]
        for k in ("fd_ext", "fd_min_atoms", "diff_substeps", "fd_n_timesteps", "fd_h", "fd_vol", "neighvol",
                  "atomic_weight", "atomic_charge", "tau", "omega_laser"):
            out["ttm_" + k] = T0[k]
        ld = np.array(T0["fd_dim"]) - 2
        gd = ld * np.array(cpu) + 2
        out["ttm_fd_dim"] = gd
        for k, v in eos.items():
            out["eos_" + k] = v
        def assemble(get):
            a0 = get(0)
            g = np.zeros(tuple(gd) + a0.shape[3:], a0.dtype)
            for rk in range(world):
                co = (rk // (cpu[1] * cpu[2]), (rk // cpu[2]) % cpu[1], rk % cpu[2])
                sl = tuple(slice(1 + co[q] * ld[q], 1 + (co[q] + 1) * ld[q]) for q in range(3))
                g[sl] = get(rk)[1:-1, 1:-1, 1:-1]
            return g
        if adv:
            out["flux_steps"] = np.array(moved)
        for s in c["dump"]:
            if adv:
                out["t%d_flux" % s] = assemble(lambda rk: T[s][rk]["stages"]["flux"]["flux"]).astype(np.int16)
                for f in ("temp", "U", "temp2", "U2", "natoms"):
                    out["t%d_adv_%s" % (s, f)] = assemble(lambda rk: T[s][rk]["stages"]["adv"]["lat"][f])
            for tag in ("update_fd", "substeps_done"):
                for f in po.TTM_REC.names:
                    if tag == "update_fd" and f not in ("natoms", "md_temp", "dens", "temp", "xi"):
                        continue
                    if tag == "substeps_done" and f not in ("natoms", "temp", "xi", "fd_k", "fd_g", "Ce", "U", "md_temp", "dens") + (("temp2", "U2") if adv else ()):
                        continue
                    a0 = T[s][0]["stages"][tag]["lat"][f]
                    g = np.zeros(tuple(gd) + a0.shape[3:], a0.dtype)
                    for rk in range(world):
                        co = (rk // (cpu[1] * cpu[2]), (rk // cpu[2]) % cpu[1], rk % cpu[2])
                        blk = T[s][rk]["stages"][tag]["lat"][f][1:-1, 1:-1, 1:-1]
                        sl = tuple(slice(1 + co[q] * ld[q], 1 + (co[q] + 1) * ld[q]) for q in range(3))
                        g[sl] = blk
                    out["t%d_%s_%s" % (s, tag, f)] = g
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    last = c["dump"][-1]
    print("%-20s %8.1f KB  ranks=%d steps=%s nbl_count(last)=%d maxF=%.3g%s" %
          (name, os.path.getsize(path) / 1024, world, c["dump"], out["s%d_nbl_count" % last], np.abs(out["s%d_kraft" % last]).max(),
           "  steps with flux=%s" % moved if adv else ""))


TMM_CASE = dict(I0=2.0e15, lam=800e-9, sigma_t=8e-15, t0=6e-15, abs_threshold=4.5)


def tmm_phase_table(eos):
    """EOS_phase_from_r_ti stand-in: phase 1 below 1000 K, 3 above 1150 K, linear in Ti between (constant along the density:
    planar inside every grid cell, so every triangulation of the grid interpolates alike)"""
    t = np.asarray(eos["t"], np.float64)
    ph = np.clip(1.0 + 2.0 * (t - 1000.0) / 150.0, 1.0, 3.0)
    return dict(x=eos["r"], y=eos["t"], z=np.repeat(ph[None, :], len(eos["r"]), axis=0))


def read_tmm_probe(path):
    """tmm.<rank>.<step>.bin of oracle/ref_probe_ttm1d.c: do_tmm's state after sub-steps 0, 1 and the last of an MD step"""
    b = open(path, "rb").read()
    assert b[:8] == b"IMDTMM1P"
    hd = np.frombuffer(b, dtype=np.float64, count=8, offset=8)
    n, nl, o, recs = int(hd[0]), int(hd[1]), 72, {}
    while True:
        sub = int(np.frombuffer(b, dtype=np.int32, count=1, offset=o)[0]); o += 4
        if sub < 0:
            break
        m = 4 * n + nl + 7
        a = np.frombuffer(b, dtype=np.float64, count=m, offset=o); o += 8 * m
        recs[sub] = dict(eps_re=a[:n], eps_im=a[n:2 * n], active=a[2 * n:3 * n], qabs=a[3 * n:4 * n], source=a[4 * n:4 * n + nl], scalars=a[4 * n + nl:])
    return dict(nx=n, nlocal=nl, nsub=int(hd[2]), k0=hd[3], omega_laser=hd[4], tmm_time=hd[5], hx=hd[6], abs_threshold=hd[7], subs=recs)


def epress_table(eos):
    """EOS_pe_from_r_te stand-in [Pa; the look-up converts to eV/A^3, imd_ttm.c:2692], planar in (rho, Te): 4.8 GPa at solid
    density and 1000 K -- a pressure whose gradient across the slab's surfaces pushes on the atoms there with a few 0.01 eV/A"""
    R, T = np.meshgrid(eos["r"], eos["t"], indexing="ij")
    return dict(x=eos["r"], y=eos["t"], z=1.9e6 * R - 9.6e8 + 6.4e5 * T)


def run_ttm1d_case(tmm=False, ep=False):
    """imd_ttm.c's own 1D TTM (the file the reference's Makefile builds): oracle/_ref/imd_mpi_eam_ttm1d_nbl on TWO ranks (its
    ttm_fill_ghost_layers and do_COMMFLUX have no one-rank branch), a slab with vacuum on both x sides, do_ADV on; the
    lattice rows of both ranks assembled into the global row, stage by stage (oracle/ref_probe_ttm1d.c), the global slab
    arrays, the virtual lattice, and the atoms of every dumped step in rank order.
    tmm: the build with the reference's own transfer-matrix laser source (imd_tmm.c; oracle/_ref/imd_mpi_eam_ttm1d_tmm_nbl):
    do_tmm at the top of every sub-step as in calc_ttm (:113), a 19 fs pulse peaking at 6 fs, on a longer slab; synthetic
    alu_eps_bb.dat / K12.dat / phase table (imd_amd/tmm.py) -> ttm1d_tmm_slab.npz with do_tmm's arrays after sub-steps 0, 1, last."""
    if ep:
        tmm = True
        name, steps, dump = "ttm1d_ep_slab", 8, [0, 1, 2, 3, 4, 8]
    elif tmm:
        name, steps, dump = "ttm1d_tmm_slab", 12, [0, 1, 2, 3, 8, 12]
    else:
        name, steps, dump = "ttm1d_slab", 20, [0, 1, 2, 8, 9, 20]
    par = COMMON.format(pbc="0 1 1", seed=12345, cpu="2 1 1").replace("ensemble nve", "ensemble ttm")
    par += ("box_param %s 6 6\nstarttemp 0.09\nmaxsteps %d\ncore_potential_file phi.pt\nembedding_energy_file F.pt\n"
            "atomic_e-density_file rho.pt\nnbl_margin 0.5\nnbl_size 1.5\nshiftx_front 8.0\nshiftx_rear 8.0\nmsgbuf_size 3.0\n"
            "fd_n_timesteps 100\nfd_min_atoms 12\natomic_weight 26.9815\natomic_charge 13\nttm_int 100000\nttmdimx %s\n"
            "vlatdim 6\nvlatbuffer 1\n" % ("24" if tmm else "12", steps, "26" if tmm else "14"))
    if tmm:
        c = TMM_CASE
        par += "I0 %r\nlambda %r\nlaser_sigma_t %r\nlaser_t_0 %r\ntmm_threshold %r\n" % (c["I0"], c["lam"], c["sigma_t"], c["t0"], c["abs_threshold"])
    else:
        par += "I0 1.0\nlambda 800e-9\nlaser_sigma_t 100e-15\nlaser_t_0 1.0\n"
    with tempfile.TemporaryDirectory() as top:
        d = os.path.join(top, "run"); os.mkdir(d)
        potential.write_synthetic_eam_files(d)
        eos = write_planar_eos(top)                       # the 1D file reads its tables from the PARENT directory (imd_ttm.c:1121-1123)
        from imd_amd import ttm as ttm_mod
        phase = tmm_phase_table(eos) if tmm else dict(x=eos["r"], y=eos["t"], z=np.ones((len(eos["r"]), len(eos["t"]))))
        ttm_mod.write_eos_table(os.path.join(top, "EOS_phase_from_r_ti.txt"), phase)
        if ep:
            ttm_mod.write_eos_table(os.path.join(top, "EOS_pe_from_r_te.txt"), epress_table(eos))
        if tmm:
            from imd_amd import tmm as tmm_mod
            tmm_mod.write_arr(os.path.join(top, "alu_eps_bb.dat"), tmm_mod.synthetic_eps_bb())
            tmm_mod.write_arr(os.path.join(top, "K12.dat"), tmm_mod.synthetic_K12())
        open(os.path.join(d, "t.param"), "w").write(par)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, dump)), IMD_PROBE_ADV="1")
        if tmm:
            env["IMD_PROBE_TMM"] = "1"
        exe = "imd_mpi_eam_ttm1d_tmm_ep_nbl" if ep else "imd_mpi_eam_ttm1d_tmm_nbl" if tmm else "imd_mpi_eam_ttm1d_nbl"
        r = subprocess.run([MPIEXEC, "-n", "2", os.path.join(REFBIN, exe), "-p", "t.param"],
                           cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:]); raise SystemExit("reference run failed for %s" % name)
        T = {s: [po.read_ttm1d_probe(os.path.join(d, "ttm1d.%d.%d.bin" % (rk, s))) for rk in (0, 1)] for s in dump}
        P = {s: [po.read_probe(os.path.join(d, "probe.%d.%d.bin" % (rk, s))) for rk in (0, 1)] for s in dump}
        M = {s: [read_tmm_probe(os.path.join(d, "tmm.%d.%d.bin" % (rk, s))) for rk in (0, 1)] for s in dump} if tmm else None
        E = {s: [np.fromfile(os.path.join(d, "epress.%d.%d.bin" % (rk, s))).reshape(2, -1) for rk in (0, 1)] for s in dump} if ep else None
    T0, P0 = T[0][0], P[0][0]
    out = dict(case=name, param_file=par, steps=np.array(dump), ref_ranks=2)
    if ep:
        out["epress_z"] = epress_table(eos)["z"]
        for s in dump:                                   # epress_global is all-reduced; every rank computes epress_deriv for ITS share of the row only (:596-620)
            assert np.array_equal(E[s][0][0], E[s][1][0])
            out["e%d_epress_global" % s] = E[s][0][0]
            out["e%d_epress_deriv_rank0" % s] = E[s][0][1]; out["e%d_epress_deriv_rank1" % s] = E[s][1][1]
    if tmm:
        from imd_amd import tmm as tmm_mod
        for k, v in TMM_CASE.items():
            out["tmm_" + k] = v
        out["tmm_eps_bb"] = tmm_mod.synthetic_eps_bb(); out["tmm_K12"] = tmm_mod.synthetic_K12(); out["tmm_phase_z"] = phase["z"]
        out["tmm_k0"] = M[dump[0]][0]["k0"]; out["tmm_omega_laser"] = M[dump[0]][0]["omega_laser"]
        for s in dump:
            m0, m1 = M[s]
            lx = m0["nlocal"]
            out["m%d_tmm_time" % s] = m0["tmm_time"]; out["m%d_nsub" % s] = m0["nsub"]
            for sub, rec in m0["subs"].items():
                q = "m%d_sub%d_" % (s, sub)
                for f in ("eps_re", "eps_im", "active", "qabs"):          # rank 0 holds what it solved (MPI_Reduce to 0)
                    out[q + f] = rec[f]
                out[q + "source"] = np.concatenate([rec["source"][1:lx - 1], m1["subs"][sub]["source"][1:lx - 1]])
                out[q + "scalars"] = rec["scalars"]                        # I_t, tmm_time, refl, trans, abs, laser_active, Eabs_local (rank 0)
                out[q + "eabs"] = rec["scalars"][6] + m1["subs"][sub]["scalars"][6]
    for k in ("ntypes", "global_cell_dim", "pbc_dirs", "box", "cellsz", "nbl_margin", "timestep", "natoms", "flags"):
        out[k] = P0[k]
    out["cell_dim"] = np.array(P0["global_cell_dim"]) + 2
    for tk in ("pair_pot", "rho_h_tab", "embed_pot"):
        for f, v in P0[tk].items():
            out["%s_%s" % (tk, f)] = v
    for k in ("fd_min_atoms", "fd_n_timesteps", "vlatbuffer", "fd_hx", "fd_hy", "fd_hz", "fd_vol", "neighvol", "atomic_weight",
              "atomic_charge", "vlatdens", "Temin", "global_fd_dim", "vlatdim"):
        out["ttm_" + k] = T0[k]
    for k, v in eos.items():
        out["eos_" + k] = v
    for s in dump:
        pre = "s%d_" % s
        cols = {k: [] for k in ("nummer", "sorte", "ort", "masse", "impuls", "kraft", "numneighs")}
        for Pr in P[s]:
            start = np.concatenate([[0], np.cumsum(Pr["cell_n"])])
            idx = np.concatenate([np.arange(start[cc], start[cc + 1]) for cc in Pr["cells"]] + [np.zeros(0, int)]).astype(int)
            cols["nummer"].append(Pr["all_nummer"][idx]); cols["sorte"].append(Pr["all_sorte"][idx]); cols["ort"].append(Pr["all_ort"][idx])
            for k in ("masse", "impuls", "kraft", "numneighs"):
                cols[k].append(Pr[k])
        for k, v in cols.items():
            a = np.concatenate(v)
            out[pre + k] = a.astype(np.int32) if k in ("nummer", "numneighs") else a.astype(np.int8) if k == "sorte" else a
        out[pre + "nbl_count"] = P[s][0]["nbl_count"]
        lx = T0["local_fd_dim"]
        for tag in T[s][0]["stages"]:
            st = [T[s][rk]["stages"][tag] for rk in (0, 1)]
            q = "t%d_%s_" % (s, tag)
            for f in po.TTM1D_CELL:                      # global row = the two ranks' inner rows
                out[q + f] = np.concatenate([st[0]["cells"][f][1:lx - 1], st[1]["cells"][f][1:lx - 1]])
            for f in po.TTM1D_GLOB:
                assert np.array_equal(st[0]["glob"][f], st[1]["glob"][f]) or f == "xiarr_global"
                out[q + f] = st[0]["glob"][f]
            vp = st[0]["vlattice_proc"]
            for f in po.TTM1D_VLAT:
                out[q + "vlat_" + f] = st[vp]["vlat"][f] if 0 <= vp < 2 else st[0]["vlat"][f]
            out[q + "scalars"] = np.array([st[0]["last_active_cell"], vp, st[0]["diff_substeps"], st[0]["max_dt_ttm"]])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB  steps=%s stages=%s last_active=%d vlat_proc=%d substeps=%d" %
          (name, os.path.getsize(path) / 1024, dump, list(T[dump[-1]][0]["stages"]), int(out["t%d_cfl_scalars" % dump[-1]][0]),
           int(out["t%d_cfl_scalars" % dump[-1]][1]), int(out["t%d_cfl_scalars" % dump[-1]][2])))


def tricub_tables():
    """Four synthetic tables over (rho, lg Te, lg Ti): nodes at the cell centres (what do_tricubinterp's cell search
    assumes), nodes from end to end of the range, unevenly spaced nodes, and a table two nodes wide."""
    from imd_amd.tricub import TricubTable
    def fun(x, y, z, s):
        X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
        return s * (1.0 + 0.3 * np.sin(0.002 * X + s) + 0.2 * Y * Y - 0.15 * Z + 0.05 * np.cos(3.0 * Y * Z) + 1e-4 * X * Z)
    tabs = {}
    rng = (1800.0, 3000.0, -1.6, 1.2, -1.6, 0.4)
    def centres(lo, hi, n): return lo + (np.arange(n) + 0.5) * (hi - lo) / n
    x, y, z = centres(rng[0], rng[1], 7), centres(rng[2], rng[3], 6), centres(rng[4], rng[5], 5)
    tabs["centred"] = TricubTable(rng, x, y, z, fun(x, y, z, 2.7))
    x, y, z = np.linspace(rng[0], rng[1], 9), np.linspace(rng[2], rng[3], 8), np.linspace(rng[4], rng[5], 6)
    tabs["end_to_end"] = TricubTable(rng, x, y, z, fun(x, y, z, 0.9))
    r = np.random.default_rng(41)
    def uneven(lo, hi, n): return np.sort(lo + (hi - lo) * r.random(n))
    x, y, z = uneven(rng[0], rng[1], 6), uneven(rng[2], rng[3], 7), uneven(rng[4], rng[5], 5)
    tabs["uneven"] = TricubTable(rng, x, y, z, fun(x, y, z, 1.4) + 0.05 * r.normal(size=(6, 7, 5)))
    x, y, z = centres(rng[0], rng[1], 2), centres(rng[2], rng[3], 3), centres(rng[4], rng[5], 1)
    tabs["narrow"] = TricubTable(rng, x, y, z, fun(x, y, z, -0.6))
    return tabs


def tricub_queries(tab, n=400, seed=3):
    """random points of the range, the ends and corners of the range, the nodes themselves, points just outside"""
    r = np.random.default_rng(seed)
    lo, hi = tab.range[0::2], tab.range[1::2]
    q = [lo + (hi - lo) * r.random((n, 3))]
    ends = np.array([[a, b, c] for a in (lo[0], hi[0], 0.5 * (lo[0] + hi[0])) for b in (lo[1], hi[1], 0.3 * lo[1] + 0.7 * hi[1])
                     for c in (lo[2], hi[2], 0.6 * lo[2] + 0.4 * hi[2])])
    q.append(ends)
    X, Y, Z = np.meshgrid(tab.x, tab.y, tab.z, indexing="ij")
    q.append(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
    out = lo + (hi - lo) * r.random((12, 3))
    out[0:2, 0] = (lo[0] - 1.0, hi[0] + 1.0); out[2:4, 1] = (lo[1] - 0.01, hi[1] + 0.01); out[4:6, 2] = (lo[2] - 0.01, hi[2] + 0.01)
    q.append(out[:6])
    return np.ascontiguousarray(np.concatenate(q))


def run_tricub_case():
    """VERDICT r2 'missing' 6: the tricubic look-up behind fitDL (imd_ttm_3D.c:2919-2968).  The reference READS each
    synthetic table with its own read_tricub_interp and evaluates its own do_tricubinterp (oracle/ref_probe_tricub.c)
    -> tests/golden/tricub.npz: tables, query points and the reference's answers (-1 outside the table)."""
    from imd_amd.tricub import write_table
    exe = os.path.join(REFBIN, "ref_probe_tricub")
    o = {}
    with tempfile.TemporaryDirectory() as d:
        for name, tab in tricub_tables().items():
            q = tricub_queries(tab)
            write_table(os.path.join(d, "t.txt"), tab)
            with open(os.path.join(d, "q.bin"), "wb") as fh:
                fh.write(np.int32(len(q)).tobytes()); fh.write(q.tobytes())
            r = subprocess.run([exe, os.path.join(d, "t.txt"), os.path.join(d, "q.bin"), os.path.join(d, "o.bin")], cwd=d,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                print(r.stdout[-3000:]); raise SystemExit("ref_probe_tricub failed")
            raw = np.fromfile(os.path.join(d, "o.bin"), dtype=np.float64)
            hd, val = raw[:12], raw[12:]
            assert len(val) == len(q) and tuple(int(v) for v in hd[:3]) == tab.dims
            for k, v in (("range", tab.range), ("x", tab.x), ("y", tab.y), ("z", tab.z), ("f", tab.f), ("queries", q), ("ref", val),
                         ("ref_header", hd)):
                o["%s_%s" % (name, k)] = v
            print("%-12s %d queries, %d outside, %d with |value| differing from every node" % (
                name, len(q), int((val == -1).sum()), int((~np.isin(val, tab.f) & (val != -1)).sum())))
    path = os.path.join(OUT, "tricub.npz")
    np.savez_compressed(path, **o)
    print("%-20s %8.1f KB" % ("tricub", os.path.getsize(path) / 1024))


def eos_triangulation_grids():
    """96 rectilinear grids: every size from 2x2 to 6x6, random sizes up to 45 x 45, the sizes of the fixtures' tables; uniform
    spacing, geometric / quadratic spacing and random node positions (the map must not depend on the coordinates)"""
    r = np.random.default_rng(11)
    sizes = [(a, b) for a in range(2, 7) for b in range(2, 7)] + [(40, 24), (32, 32), (3, 60), (60, 3)]
    sizes += [(9, 11)] * 3 + [(24, 40)] * 3 + [(13, 17)] * 3              # the same size with all three kinds of coordinates
    base = len(sizes)
    sizes += [(int(a), int(b)) for a, b in r.integers(2, 46, size=(96 - len(sizes), 2))]
    out = []
    for q, (nx, ny) in enumerate(sizes):
        kind = (q - 29) % 3 if 29 <= q < base else q % 3
        if kind == 0:
            x, y = np.linspace(500.0, 4500.0, nx), np.linspace(1.0, 245.0, ny)
        elif kind == 1:
            x, y = 500.0 * 1.17 ** np.arange(nx), np.linspace(1.0, 245.0, ny) ** 2 / 50.0
        else:
            x, y = np.sort(r.uniform(0.0, 100.0, nx)), np.sort(r.uniform(0.0, 3.0, ny))
        out.append((nx, ny, x, y))
    return out


def run_eos_triangulation_case():
    """which diagonal the reference's triangulator (nn-c / Triangle inside nn_interpol/libnn.a) puts into every
    cell of a table grid, asked of the archive directly (oracle/ref_probe_tri.c) -> tests/golden/eos_triangulation.npz"""
    exe = os.path.join(REFBIN, "ref_probe_tri")
    o = {}
    with tempfile.TemporaryDirectory() as d:
        for q, (nx, ny, x, y) in enumerate(eos_triangulation_grids()):
            with open(os.path.join(d, "g.txt"), "w") as fh:
                fh.write("%d %d\n%s\n%s\n" % (nx, ny, " ".join("%.17g" % v for v in x), " ".join("%.17g" % v for v in y)))
            r = subprocess.run([exe, os.path.join(d, "g.txt")], stdout=subprocess.PIPE, text=True)
            rows = r.stdout.split()
            assert r.returncode == 0 and len(rows) == nx - 1 and all(len(w) == ny - 1 for w in rows) and "?" not in r.stdout
            o["g%d_x" % q] = x; o["g%d_y" % q] = y
            o["g%d_diag" % q] = np.array([[1 if c == "/" else 2 for c in w] for w in rows], np.uint8)
        o["n"] = q + 1
    path = os.path.join(OUT, "eos_triangulation.npz")
    np.savez_compressed(path, **o)
    print("%-20s %8.1f KB  %d grids" % ("eos_triangulation", os.path.getsize(path) / 1024, int(o["n"])))


def run_chkpt_case():
    """Configuration-file fixture (imd_amd/config_io.py) -> tests/golden/chkpt.npz:
    run A writes an ASCII checkpoint after step 10, run B the same state as a binary one (binary_output 1); then
    imd_amd.config_io.write_config writes B's state back as a binary file and the reference READS it (run C,
    coordname mine.chkpt, box from the header): its memory at step 0 is stored as reread_*."""
    from imd_amd import config_io
    c = dict(binary="imd_eam_nve_nbl", box="4 4 4", temp=0.06, steps=11, dump=[10, 11], pot="eam", nbl=True,
             extra="checkpt_int 10\n")
    files, probes = {}, {}
    for tag, extra in (("A", ""), ("L", "binary_output 1\n")):
        with tempfile.TemporaryDirectory() as d:
            par, _ = write_inputs(dict(c, extra=c["extra"] + extra), d)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="10,11")
            r = subprocess.run([os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                print(r.stdout[-3000:]); raise SystemExit("reference run failed")
            files[tag] = open(os.path.join(d, "out.00001.chkpt"), "rb").read()
            probes[tag] = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in (10, 11)}
    out = dict(chkpt_A=np.frombuffer(files["A"], np.uint8), chkpt_L=np.frombuffer(files["L"], np.uint8))
    P10, P11 = probes["L"][10], probes["L"][11]
    for s, P in ((10, P10), (11, P11)):
        start = np.concatenate([[0], np.cumsum(P["cell_n"])])
        idx = np.concatenate([np.arange(start[cc], start[cc + 1]) for cc in P["cells"]])
        out["s%d_nummer" % s] = P["all_nummer"][idx]; out["s%d_ort" % s] = P["all_ort"][idx]
        for k in ("masse", "impuls", "pot_eng", "eam_rho"):
            out["s%d_%s" % (s, k)] = P[k]
    out["box"] = P10["box"]; out["pbc_dirs"] = P10["pbc_dirs"]
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "L.chkpt"), "wb").write(files["L"])
        st = config_io.read_config(os.path.join(d, "L.chkpt"), ntypes=1)
        config_io.write_config(os.path.join(d, "mine.chkpt"), st["number"], st["vtype"], st["mass"], st["pos"], st["impuls"],
                               st["data"]["Epot"], st["header"]["box"], st["header"]["pbc"], eam_rho=st["data"]["eam_rho"],
                               binary=True, stamp="(fixture)")
        out["mine_L"] = np.frombuffer(open(os.path.join(d, "mine.chkpt"), "rb").read(), np.uint8)
        par, _ = write_inputs(dict(c, steps=0, extra=""), d)
        par = par.replace("coordname _fcc", "coordname mine.chkpt").replace("box_param 4 4 4\n", "box_from_header 1\n")
        open(os.path.join(d, "t.param"), "w").write(par)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0")
        r = subprocess.run([os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:]); raise SystemExit("reference re-read run failed")
        P = po.read_probe(os.path.join(d, "probe.0.0.bin"))
        start = np.concatenate([[0], np.cumsum(P["cell_n"])])
        idx = np.concatenate([np.arange(start[cc], start[cc + 1]) for cc in P["cells"]])
        out["reread_nummer"] = P["all_nummer"][idx]; out["reread_ort"] = P["all_ort"][idx]
        out["reread_impuls"] = P["impuls"]; out["reread_masse"] = P["masse"]; out["reread_box"] = P["box"]
        out["reread_param_file"] = par
    path = os.path.join(OUT, "chkpt.npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB  %d atoms" % ("chkpt", os.path.getsize(path) / 1024, len(out["s11_nummer"])))


def run_ttm_file_case():
    """`.ttm` lattice file (imd_amd/ttm_io.py) -> tests/golden/ttm_file.npz: the file ttm_writeout wrote after step 2
    (ttm_int 1) and the lattice the reference held at the end of calc_ttm of that step (probe stage `cfl`)."""
    c = dict(CASES["ttm_2048"], steps=2, dump=[2])
    with tempfile.TemporaryDirectory() as d:
        par, _ = write_inputs(c, d)
        par = par.replace("ttm_int 100000", "ttm_int 1")
        open(os.path.join(d, "t.param"), "w").write(par)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="2")
        r = subprocess.run([os.path.join(REFBIN, c["binary"]), "-p", "t.param"], cwd=d, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:]); raise SystemExit("reference run failed")
        text = open(os.path.join(d, "out.2.ttm"), "rb").read()
        T = po.read_ttm_probe(os.path.join(d, "ttm.0.2.bin"))
    out = dict(ttm_file=np.frombuffer(text, np.uint8), param_file=par)
    for f in po.TTM_REC.names:
        out["lat_" + f] = T["stages"]["cfl"]["lat"][f]
    path = os.path.join(OUT, "ttm_file.npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB" % ("ttm_file", os.path.getsize(path) / 1024))


if __name__ == "__main__":
    names = sys.argv[1:] or (list(CASES) + ["eos_inverse", "eos_grid", "fdtd_1d", "chkpt", "ttm_file", "ttm1d_slab", "ttm1d_tmm_slab", "ttm1d_ep_slab", "tricub", "eos_triangulation"])
    for n in names:
        if n == "eos_inverse":
            run_eos_inverse()
        elif n == "eos_grid":
            run_eos_grid()
        elif n == "fdtd_1d":
            run_fdtd_case()
        elif n == "ttm_file":
            run_ttm_file_case()
        elif n == "chkpt":
            run_chkpt_case()
        elif n == "ttm_adv":
            run_adv_case(n, CASES[n])
        elif n == "ttm1d_slab":
            run_ttm1d_case()
        elif n == "ttm1d_tmm_slab":
            run_ttm1d_case(tmm=True)
        elif n == "ttm1d_ep_slab":
            run_ttm1d_case(ep=True)
        elif n == "eos_triangulation":
            run_eos_triangulation_case()
        elif n == "tricub":
            run_tricub_case()
        elif n.endswith("_mpi8") or n.endswith("_mpi2"):
            run_mpi_case(n, CASES[n])
        else:
            run_case(n, CASES[n])
