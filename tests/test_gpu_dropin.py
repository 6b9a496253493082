"""Drop-in integration test (run with -m gpu): the REAL reference program, built here from /root/reference
with its imd_forces_nbl.o replaced by imd_amd/host/imd_shim.c + libimdgpu.so (oracle/build_ref.sh dropin),
runs the same input as the unmodified CPU reference binary.  The reference's own main loop, integrator,
fix_cells and parameter/potential readers drive the GPU forces; trajectories must agree."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from imd_amd import potential
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = os.path.join(ROOT, "oracle", "_ref", "imd_eam_nve_nbl")
GPU = os.path.join(ROOT, "oracle", "_ref", "imd_eam_nve_nbl_gpu")
CPU_CP = os.path.join(ROOT, "oracle", "_ref", "imd_eam_nve")           # cell-pair force loop (no -DNBL)
GPU_CP = os.path.join(ROOT, "oracle", "_ref", "imd_eam_nve_gpu")

PARAM = """coordname _fcc
outfiles out
ensemble nve
maxsteps 30
timestep 0.1
ntypes 1
total_types 1
masses 26.9815
types 0
box_param 10 10 10
box_unit 4.05
pbc_dirs 1 1 1
starttemp 0.06
seed 4711
core_potential_file phi.pt
embedding_energy_file F.pt
atomic_e-density_file rho.pt
eng_int 1
checkpt_int 100000
nbl_margin 0.5
nbl_size 1.5
cpu_dim 1 1 1
"""


@pytest.mark.skipif(not (os.path.exists(CPU) and os.path.exists(GPU)), reason="oracle/_ref binaries not built")
def test_reference_main_loop_on_gpu_forces():
    out = {}
    for tag, exe in (("cpu", CPU), ("gpu", GPU)):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            open(os.path.join(d, "t.param"), "w").write(PARAM)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0,1,30")
            r = subprocess.run([exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-2000:]
            out[tag] = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in (0, 1, 30)}
            out[tag]["eng"] = np.loadtxt(os.path.join(d, "out.eng"))
    for s in (0, 1, 30):
        a, b = out["gpu"][s], out["cpu"][s]
        oa, ob = np.argsort(_inner_ids(a)), np.argsort(_inner_ids(b))
        fa, fb = a["kraft"][oa], b["kraft"][ob]
        assert np.abs(fa - fb).max() < 1e-9 * max(np.abs(fb).max(), 1e-3), "forces differ at step %d" % s
        assert np.abs(a["pot_eng"][oa] - b["pot_eng"][ob]).max() < 1e-10
        assert np.abs(a["eam_rho"][oa] - b["eam_rho"][ob]).max() < 1e-10
        assert abs(a["tot_pot_energy"] - b["tot_pot_energy"]) < 1e-9 * abs(b["tot_pot_energy"])
        assert abs(a["virial"] - b["virial"]) < 1e-9 * abs(b["virial"])
        pa, pb = _inner_pos(a)[oa], _inner_pos(b)[ob]
        assert np.abs(pa - pb).max() < 1e-9
    # the shim publishes the list in the reference's tl/tb/cl_off globals (globals.h:184-188): bit-exact at step 0
    for key in ("cl_off", "tl", "tb"):
        assert np.array_equal(out["gpu"][0][key], out["cpu"][0][key]), key
    assert out["gpu"][30]["nbl_count"] == out["cpu"][30]["nbl_count"] >= 2     # same rebuild schedule
    assert np.allclose(out["gpu"]["eng"], out["cpu"]["eng"], rtol=1e-6, atol=1e-9)   # .eng is printed with ~6-8 digits


def _inner_ids(P):
    start = np.concatenate([[0], np.cumsum(P["cell_n"])])
    return np.concatenate([P["all_nummer"][start[c]:start[c + 1]] for c in P["cells"]])


def _inner_pos(P):
    start = np.concatenate([[0], np.cumsum(P["cell_n"])])
    return np.concatenate([P["all_ort"][start[c]:start[c + 1]] for c in P["cells"]])


@pytest.mark.skipif(not (os.path.exists(CPU_CP) and os.path.exists(GPU_CP)), reason="oracle/_ref binaries not built")
def test_reference_cellpair_main_loop_on_gpu_forces():
    """builds without -DNBL enter through calc_forces -> do_forces / do_forces_eam2
    (imd_main_risc_3d.c:31, imd_forces.c:33, imd_forces_eam2.c:239).  Same program, shim instead of that loop."""
    param = "\n".join(l for l in PARAM.splitlines() if not l.startswith("nbl_")) + "\n"
    out = {}
    for tag, exe in (("cpu", CPU_CP), ("gpu", GPU_CP)):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            open(os.path.join(d, "t.param"), "w").write(param)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0,1,30")
            r = subprocess.run([exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-2000:]
            out[tag] = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in (0, 1, 30)}
            out[tag]["eng"] = np.loadtxt(os.path.join(d, "out.eng"))
    for s in (0, 1, 30):
        a, b = out["gpu"][s], out["cpu"][s]
        oa, ob = np.argsort(_inner_ids(a)), np.argsort(_inner_ids(b))
        fa, fb = a["kraft"][oa], b["kraft"][ob]
        assert np.abs(fa - fb).max() < 1e-9 * max(np.abs(fb).max(), 1e-3), "forces differ at step %d" % s
        assert np.abs(a["pot_eng"][oa] - b["pot_eng"][ob]).max() < 1e-10
        assert np.abs(a["eam_rho"][oa] - b["eam_rho"][ob]).max() < 1e-10
        assert abs(a["tot_pot_energy"] - b["tot_pot_energy"]) < 1e-9 * abs(b["tot_pot_energy"])
        assert abs(a["virial"] - b["virial"]) < 1e-9 * abs(b["virial"])
        assert np.abs(_inner_pos(a)[oa] - _inner_pos(b)[ob]).max() < 1e-9
    assert np.allclose(out["gpu"]["eng"], out["cpu"]["eng"], rtol=1e-6, atol=1e-9)


CPU_TTM = os.path.join(ROOT, "oracle", "_ref", "imd_mpi_eam_ttm3d_nbl")
GPU_TTM = os.path.join(ROOT, "oracle", "_ref", "imd_mpi_eam_ttm3d_nbl_gpu")

PARAM_TTM = PARAM.replace("ensemble nve", "ensemble ttm").replace("box_param 10 10 10", "box_param 8 8 8") \
    .replace("starttemp 0.06", "starttemp 0.03").replace("maxsteps 30", "maxsteps 12") + """fd_ext 1 1 1
fd_n_timesteps 100
fd_min_atoms 5
atomic_weight 26.9815
atomic_charge 13
ttm_int 6
I0 1.0
lambda 800e-9
laser_sigma_t 100e-15
laser_t_0 1.0
"""


def _write_planar_eos(d):
    """the synthetic EOS tables of the fixtures (tests/golden/make_golden.py:write_planar_eos)"""
    nr, nt = 24, 40
    r = np.linspace(500.0, 4500.0, nr); t = np.linspace(1.0, 60000.0, nt); ts = np.linspace(1.0, 245.0, nt)
    zc = np.outer(np.ones(nr), 0.05 * t); ze = np.outer(np.ones(nr), 4000.0 * ts)
    for fn, y, z in (("EOS_cve_from_r_te.txt", t, zc), ("EOS_ee_from_r_tesqrt.txt", ts, ze)):
        with open(os.path.join(d, fn), "w") as f:
            f.write("%d %d\n%.16e %.16e %.16e %.16e\n" % (nr, nt, r[0], r[-1], y[0], y[-1]))
            for i in range(nr):
                for j in range(nt):
                    f.write("%.16e %.16e %.16e\n" % (r[i], y[j], z[i, j]))


def _write_curved_eos(d):
    """curved, non-separable tables: here the reference's triangle-linear interpolation (nn-c) and a bilinear rule differ
    inside every grid cell -- the shim hands the library the reference's own triangulation (imd_shim.c:eos_triangulation)"""
    nr, nt = 24, 40
    r = np.linspace(500.0, 4500.0, nr); t = np.linspace(1.0, 60000.0, nt); ts = np.linspace(1.0, 245.0, nt)
    R, T = np.meshgrid(r, t, indexing="ij")
    zc = 0.05 * T * (1.0 + 0.25 * np.sin(R / 700.0)) * (1.0 + 0.5 * T / 60000.0) + 2.0
    R, S = np.meshgrid(r, ts, indexing="ij")
    ze = 4000.0 * S * (1.0 + 0.15 * np.cos(R / 900.0)) + 6.0 * S * S
    for fn, y, z in (("EOS_cve_from_r_te.txt", t, zc), ("EOS_ee_from_r_tesqrt.txt", ts, ze)):
        with open(os.path.join(d, fn), "w") as f:
            f.write("%d %d\n%.16e %.16e %.16e %.16e\n" % (nr, nt, r[0], r[-1], y[0], y[-1]))
            for i in range(nr):
                for j in range(nt):
                    f.write("%.16e %.16e %.16e\n" % (r[i], y[j], z[i, j]))


@pytest.mark.skipif(not (os.path.exists(CPU_TTM) and os.path.exists(GPU_TTM)), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("eos", ["planar", "curved"])
def test_reference_ttm_main_loop_on_gpu_forces_and_gpu_calc_ttm(eos):
    """The reference's 3D TTM program (MPI build, one rank) with BOTH hot-path entry points replaced by the shim:
    calc_forces and calc_ttm run on the device, the host keeps init_ttm (EOS tables, l1/l2), move_atoms_ttm,
    fix_cells, ttm_writeout.  Against the CPU build of the same program (its calc_ttm = the reference's own functions
    in the order of the first block, see oracle/ref_probe_ttm.c) over 12 steps: lattice (what move_atoms_ttm
    reads: xi, vcom, natoms; and temp, U, md_temp, dens, fd_k, fd_g, Ce) within 1e-9, atoms within 1e-9, and
    the .ttm file the host wrote from l1 equal line by line at its printed precision.
    eos = "curved": non-separable EOS tables, where the result depends on which triangles nn-c interpolates on -- the
    shim passes the reference's own triangulation to the library, so row a14 holds on real (curved) tables too."""
    out = {}
    steps = (0, 1, 6, 12)
    for tag, exe in (("cpu", CPU_TTM), ("gpu", GPU_TTM)):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            (_write_planar_eos if eos == "planar" else _write_curved_eos)(d)
            open(os.path.join(d, "t.param"), "w").write(PARAM_TTM)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, steps)))
            r = subprocess.run([exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-3000:]
            out[tag] = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in steps}
            out[tag]["lat"] = {s: po.read_ttm_probe(os.path.join(d, "ttm.0.%d.bin" % s))["stages"]["cfl"] for s in steps}
            out[tag]["ttm_file"] = np.loadtxt(os.path.join(d, "out.2.ttm"))
    inner = (slice(1, -1),) * 3
    for s in steps:
        a, b = out["gpu"]["lat"][s], out["cpu"]["lat"][s]
        assert np.array_equal(a["lat"]["natoms"][inner], b["lat"]["natoms"][inner])
        assert np.array_equal(a["lat"]["natoms_old"][inner], b["lat"]["natoms_old"][inner])
        for f in ("temp", "temp2", "U", "U2", "xi", "md_temp", "dens", "vcom", "fd_k", "fd_g", "Ce", "Z", "ne"):
            x, y = a["lat"][f][inner], b["lat"][f][inner]
            assert np.abs(x - y).max() <= 1e-9 * np.abs(y).max(), (s, f)
        assert abs(a["max_dt_ttm"] - b["max_dt_ttm"]) <= 1e-9 * abs(b["max_dt_ttm"])
        A, B = out["gpu"][s], out["cpu"][s]
        oa, ob = np.argsort(_inner_ids(A)), np.argsort(_inner_ids(B))
        assert np.abs(_inner_pos(A)[oa] - _inner_pos(B)[ob]).max() < 1e-9, s
        assert np.abs(A["impuls"][oa] - B["impuls"][ob]).max() < 1e-9 * np.abs(B["impuls"]).max(), s
        assert np.array_equal(A["numneighs"][oa], B["numneighs"][ob])
    assert np.allclose(out["gpu"]["ttm_file"], out["cpu"]["ttm_file"], rtol=2e-6, atol=1e-12)
    if eos == "curved":
        # not vacuous: the same run with the library's stand-alone bilinear rule (no triangulation map) is visibly different
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            _write_curved_eos(d)
            open(os.path.join(d, "t.param"), "w").write(PARAM_TTM)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="12", IMD_SHIM_EOS_BILINEAR="1")
            r = subprocess.run([GPU_TTM, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-3000:]
            bl = po.read_ttm_probe(os.path.join(d, "ttm.0.12.bin"))["stages"]["cfl"]
        x, y = bl["lat"]["temp"][inner], out["cpu"]["lat"][12]["lat"]["temp"][inner]
        assert np.abs(x - y).max() > 1e-7 * np.abs(y).max()


GPU_TTM_FN = os.path.join(ROOT, "oracle", "_ref", "imd_mpi_eam_ttm3d_nbl_gpufn")
GPU_CP_FN = os.path.join(ROOT, "oracle", "_ref", "imd_eam_nve_gpufn")


@pytest.mark.skipif(not (os.path.exists(CPU_CP) and os.path.exists(GPU_CP_FN)), reason="oracle/_ref binaries not built")
def test_reference_cell_pair_loop_calls_do_forces_on_the_device():
    """do_forces, do_forces_eam2 and do_embedding_energy under the reference's own names and signatures
    (prototypes.h:412-418): in this build the reference's OWN calc_forces (imd_main_risc_3d.c:31-182, unchanged) walks the
    cell pairs and calls them -- imd_forces.o / imd_forces_eam2.o are not linked, the shim's versions send every pair of
    cells to the device (imdgpu_do_forces / _do_forces_eam2 / _do_embedding_energy) and add the results on the host.
    Forces, energies, densities and the trajectory over 3 steps equal the CPU build's."""
    out = {}
    par = PARAM.replace("box_param 10 10 10", "box_param 6 6 6").replace("maxsteps 30", "maxsteps 3") \
        .replace("nbl_margin 0.5\n", "").replace("nbl_size 1.5\n", "")
    for tag, exe in (("cpu", CPU_CP), ("gpu", GPU_CP_FN)):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            open(os.path.join(d, "t.param"), "w").write(par)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0,3")
            r = subprocess.run([exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-2000:]
            out[tag] = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in (0, 3)}
    for s in (0, 3):
        a, b = out["gpu"][s], out["cpu"][s]
        oa, ob = np.argsort(_inner_ids(a)), np.argsort(_inner_ids(b))
        assert np.abs(a["kraft"][oa] - b["kraft"][ob]).max() < 1e-9 * max(np.abs(b["kraft"]).max(), 1e-3), s
        assert np.abs(a["pot_eng"][oa] - b["pot_eng"][ob]).max() < 1e-10
        assert np.abs(a["eam_rho"][oa] - b["eam_rho"][ob]).max() < 1e-10
        assert np.abs(a["eam_dF"][oa] - b["eam_dF"][ob]).max() < 1e-10
        assert abs(a["tot_pot_energy"] - b["tot_pot_energy"]) < 1e-10 * abs(b["tot_pot_energy"])
        assert abs(a["virial"] - b["virial"]) < 1e-9 * abs(b["virial"])
        assert np.abs(_inner_pos(a)[oa] - _inner_pos(b)[ob]).max() < 1e-9


@pytest.mark.skipif(not (os.path.exists(CPU) and os.path.exists(GPU)), reason="oracle/_ref binaries not built")
def test_level2_reference_main_loop_with_the_atoms_on_the_device():
    """Level 2 (IMD_SHIM_LEVEL=2): the reference's own main loop, but calc_forces only enqueues the device step
    (imdgpu_md_step_begin), the global `move_atoms` pointer (globals.h:1777) is re-pointed to the device integrator
    (imdgpu_md_step_end) and check_nblist reads the device's flag; the host's cells are refreshed only where the host reads
    them (IMD_SHIM_SYNC_STEPS = the probe's steps, the final checkpoint).  Trajectory, energies file and rebuild schedule
    equal the CPU binary's over 30 steps; the checkpoint written at the end holds the device's atoms."""
    out = {}
    for tag, exe, extra in (("cpu", CPU, {}), ("gpu", GPU, dict(IMD_SHIM_LEVEL="2", IMD_SHIM_SYNC_STEPS="0,1,30"))):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            open(os.path.join(d, "t.param"), "w").write(PARAM.replace("checkpt_int 100000", "checkpt_int 30"))
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0,1,30", **extra)
            r = subprocess.run([exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-2000:]
            out[tag] = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in (0, 1, 30)}
            out[tag]["eng"] = np.loadtxt(os.path.join(d, "out.eng"))
            out[tag]["chkpt"] = open(os.path.join(d, "out.00001.chkpt")).read().splitlines()
    for s in (0, 1, 30):
        a, b = out["gpu"][s], out["cpu"][s]
        oa, ob = np.argsort(_inner_ids(a)), np.argsort(_inner_ids(b))
        assert len(oa) == len(ob)
        assert np.abs(a["kraft"][oa] - b["kraft"][ob]).max() < 1e-9 * max(np.abs(b["kraft"]).max(), 1e-3), "forces differ at step %d" % s
        assert np.abs(_inner_pos(a)[oa] - _inner_pos(b)[ob]).max() < 1e-9
        assert np.abs(a["impuls"][oa] - b["impuls"][ob]).max() < 1e-9 * np.abs(b["impuls"]).max()
        assert abs(a["tot_pot_energy"] - b["tot_pot_energy"]) < 1e-9 * abs(b["tot_pot_energy"])
    assert out["gpu"][30]["nbl_count"] == out["cpu"][30]["nbl_count"] >= 2
    assert np.allclose(out["gpu"]["eng"], out["cpu"]["eng"], rtol=1e-6, atol=1e-9)
    # the checkpoint of step 30 (write_config after move_atoms): same atoms, positions to the file's printed precision
    def atoms(lines):
        rows = np.array([[float(v) for v in l.split()] for l in lines if l and not l.startswith("#")])
        return rows[np.argsort(rows[:, 0])]
    A, B = atoms(out["gpu"]["chkpt"]), atoms(out["cpu"]["chkpt"])
    assert A.shape == B.shape and np.array_equal(A[:, 0], B[:, 0])
    assert np.allclose(A[:, 3:6], B[:, 3:6], rtol=0, atol=2e-6)


@pytest.mark.skipif(not (os.path.exists(CPU) and os.path.exists(GPU)), reason="oracle/_ref binaries not built")
def test_level2_a_stop_file_writes_the_atoms_the_device_holds():
    """the reference's main loop also writes atoms outside the periodic steps -- check_stop / check_walltime call
    write_config(-1, steps) and end the run (imd_main_3d.c:1964-2007), check_write answers a `write` file (:838-843).  At level 2
    the host cells are stale between refreshes; the shim's writer wrappers (IMD_SHIM_WRAP_WRITERS, ld --wrap=write_config ...)
    hand every writer the device's atoms first.  A run that finds a stop file after its first step: the file it leaves equals
    the CPU binary's -- atoms AFTER that step's integrator, not the ones uploaded."""
    out = {}
    for tag, exe, extra in (("cpu", CPU, {}), ("gpu", GPU, dict(IMD_SHIM_LEVEL="2"))):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            open(os.path.join(d, "t.param"), "w").write(PARAM + "stop_int 1\n")
            open(os.path.join(d, "stop"), "w").write("")
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="99", **extra)      # the probe stays out of this run
            r = subprocess.run([exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
            assert r.returncode == 0 and "Stop file found after 0 steps" in r.stdout, r.stdout[-2000:]
            out[tag] = open(os.path.join(d, "out-final.chkpt")).read().splitlines()

    def atoms(lines):
        rows = np.array([[float(v) for v in l.split()] for l in lines if l and not l.startswith("#")])
        return rows[np.argsort(rows[:, 0])]
    A, B = atoms(out["gpu"]), atoms(out["cpu"])
    assert A.shape == B.shape and np.array_equal(A[:, 0], B[:, 0])
    assert np.allclose(A[:, 3:6], B[:, 3:6], rtol=0, atol=2e-6)
    # ... and those are not the uploaded positions: the first step moved the atoms by far more than the file's precision
    from imd_amd import lattice
    p0 = lattice.fcc_positions((10, 10, 10))
    moved = np.abs((A[:, 3:6] / (4.05 / 2) - 0.5) - np.round(A[:, 3:6] / (4.05 / 2) - 0.5)).max() * (4.05 / 2)
    assert moved > 1e-3 and len(p0) == len(A)


@pytest.mark.skipif(not (os.path.exists(CPU_TTM) and os.path.exists(GPU_TTM)), reason="oracle/_ref binaries not built")
def test_level2_reference_ttm_main_loop_with_atoms_and_lattice_on_the_device():
    """The 3D-TTM program at level 2: calc_forces forks calc_ttm beside the force passes on the device, move_atoms_ttm runs
    there too; l1/l2 and the cells come back on the probe's steps, on ttm_int steps (ttm_writeout) and at the end.
    Lattice within 1e-9, atoms within 1e-9, the .ttm files at their printed precision."""
    out = {}
    steps = (0, 1, 6, 12)
    for tag, exe, extra in (("cpu", CPU_TTM, {}), ("gpu", GPU_TTM, dict(IMD_SHIM_LEVEL="2", IMD_SHIM_SYNC_STEPS="0,1,6,12"))):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            _write_planar_eos(d)
            open(os.path.join(d, "t.param"), "w").write(PARAM_TTM)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, steps)), **extra)
            r = subprocess.run([exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-3000:]
            out[tag] = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in steps}
            out[tag]["lat"] = {s: po.read_ttm_probe(os.path.join(d, "ttm.0.%d.bin" % s))["stages"]["cfl"] for s in steps}
            out[tag]["ttm_file"] = [np.loadtxt(os.path.join(d, "out.%d.ttm" % k)) for k in (1, 2)]
    inner = (slice(1, -1),) * 3
    for s in steps:
        a, b = out["gpu"]["lat"][s], out["cpu"]["lat"][s]
        assert np.array_equal(a["lat"]["natoms"][inner], b["lat"]["natoms"][inner])
        for f in ("temp", "temp2", "U", "U2", "xi", "md_temp", "dens", "vcom", "fd_k", "fd_g", "Ce"):
            x, y = a["lat"][f][inner], b["lat"][f][inner]
            assert np.abs(x - y).max() <= 1e-9 * np.abs(y).max(), (s, f)
        A, B = out["gpu"][s], out["cpu"][s]
        oa, ob = np.argsort(_inner_ids(A)), np.argsort(_inner_ids(B))
        assert np.abs(_inner_pos(A)[oa] - _inner_pos(B)[ob]).max() < 1e-9, s
        assert np.abs(A["impuls"][oa] - B["impuls"][ob]).max() < 1e-9 * np.abs(B["impuls"]).max(), s
        assert np.array_equal(A["numneighs"][oa], B["numneighs"][ob])
    for x, y in zip(out["gpu"]["ttm_file"], out["cpu"]["ttm_file"]):
        assert np.allclose(x, y, rtol=2e-6, atol=1e-12)


@pytest.mark.skipif(not (os.path.exists(CPU_TTM) and os.path.exists(GPU_TTM_FN)), reason="oracle/_ref binaries not built")
def test_reference_calls_the_lattice_functions_one_by_one():
    """The per-function boundary: update_fd, do_cell_activation, do_FILLMESH, ttm_fill_ghost_layers,
    do_DIFF and CFL_maxdt exist in the shim under the reference's own names and signatures (prototypes.h:141-153,1018-1021).
    In this build of the reference program imd_ttm_3D.c's definitions of these names are renamed at compile time, so the C
    host sequence of oracle/ref_probe_ttm.c (calc_ttm's first block, one call per function, l1/l2 read by the host between
    them) reaches the DEVICE through them: 1 + 1 + 1 + 1 + 100 x (1 + 1 + 1) + 1 calls per MD step, each
    l1/l2 -> device -> call -> l1/l2.  Stage by stage the host's lattice equals the CPU build's, whose probe drives the
    reference's own functions in the same order."""
    out = {}
    steps = (0, 1, 4)
    par = PARAM_TTM.replace("maxsteps 12", "maxsteps 4")
    for tag, exe in (("cpu", CPU_TTM), ("gpu", GPU_TTM_FN)):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            _write_planar_eos(d)
            open(os.path.join(d, "t.param"), "w").write(par)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, steps)))
            r = subprocess.run([exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-3000:]
            out[tag] = {s: po.read_ttm_probe(os.path.join(d, "ttm.0.%d.bin" % s))["stages"] for s in steps}
            out[tag]["atoms"] = po.read_probe(os.path.join(d, "probe.0.4.bin"))
    inner = (slice(1, -1),) * 3
    for s in steps:
        for stage in ("update_fd", "fillmesh0", "diff_only1", "substep1", "substep2", "substeps_done", "cfl"):
            a, b = out["gpu"][s][stage], out["cpu"][s][stage]
            assert np.array_equal(a["lat"]["natoms"][inner], b["lat"]["natoms"][inner]), (s, stage)
            for f in ("temp", "temp2", "U", "U2", "xi", "md_temp", "dens", "fd_k", "fd_g", "Ce"):
                x, y = a["lat"][f][inner], b["lat"][f][inner]
                assert np.abs(x - y).max() <= 1e-9 * max(np.abs(y).max(), 1e-300), (s, stage, f)
        assert abs(out["gpu"][s]["cfl"]["max_dt_ttm"] - out["cpu"][s]["cfl"]["max_dt_ttm"]) <= 1e-9 * abs(out["cpu"][s]["cfl"]["max_dt_ttm"])
    A, B = out["gpu"]["atoms"], out["cpu"]["atoms"]
    oa, ob = np.argsort(_inner_ids(A)), np.argsort(_inner_ids(B))
    assert np.abs(_inner_pos(A)[oa] - _inner_pos(B)[ob]).max() < 1e-9


@pytest.mark.skipif(not (os.path.exists(CPU_TTM) and os.path.exists(GPU_TTM)), reason="oracle/_ref binaries not built")
def test_reference_ttm_restart_keeps_the_electron_state_on_gpu():
    """a restarted run (`-r 1`: init_ttm -> ttm_read fills l1/l2 from out.1.ttm, imd_ttm_3D.c:1284-1290) must
    continue from the checkpointed electron lattice on the device too.  The CPU build writes checkpoint + .ttm file at
    step 6; both builds are restarted from those files and run steps 7..12; lattices and atoms must agree (the
    shim uploads the host lattice once, at the first calc_ttm -- without that the device would start from zeros)."""
    import shutil
    steps = (7, 12)
    with tempfile.TemporaryDirectory() as d0:
        potential.write_synthetic_eam_files(d0)
        _write_planar_eos(d0)
        par = PARAM_TTM.replace("checkpt_int 100000", "checkpt_int 6")
        open(os.path.join(d0, "t.param"), "w").write(par.replace("maxsteps 12", "maxsteps 6"))
        r = subprocess.run([CPU_TTM, "-p", "t.param"], cwd=d0, env=dict(os.environ, IMD_PROBE_STEPS="-1"),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and os.path.exists(os.path.join(d0, "out.1.ttm")), r.stdout[-2000:]
        open(os.path.join(d0, "t.param"), "w").write(par)
        out = {}
        for tag, exe in (("cpu", CPU_TTM), ("gpu", GPU_TTM)):
            d = os.path.join(d0, tag)
            shutil.copytree(d0, d, ignore=shutil.ignore_patterns("cpu", "gpu"))
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, steps)))
            r = subprocess.run([exe, "-r", "1", "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=600)
            assert r.returncode == 0 and "Reading 64 lines from ttm-file" in r.stdout, r.stdout[-3000:]
            out[tag] = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in steps}
            out[tag]["lat"] = {s: po.read_ttm_probe(os.path.join(d, "ttm.0.%d.bin" % s))["stages"]["cfl"] for s in steps}
    inner = (slice(1, -1),) * 3
    for s in steps:
        a, b = out["gpu"]["lat"][s], out["cpu"]["lat"][s]
        assert np.array_equal(a["lat"]["natoms"][inner], b["lat"]["natoms"][inner])
        # the electrons are NOT in equilibrium with the lattice in this state: temp carries the checkpointed history
        assert np.abs(b["lat"]["temp"][inner] - b["lat"]["md_temp"][inner]).max() > 1e-6
        for f in ("temp", "U", "xi", "md_temp", "dens", "vcom", "fd_k", "fd_g", "Ce"):
            x, y = a["lat"][f][inner], b["lat"][f][inner]
            assert np.abs(x - y).max() <= 1e-9 * np.abs(y).max(), (s, f)
        A, B = out["gpu"][s], out["cpu"][s]
        oa, ob = np.argsort(_inner_ids(A)), np.argsort(_inner_ids(B))
        assert np.abs(_inner_pos(A)[oa] - _inner_pos(B)[ob]).max() < 1e-9, s
        assert np.abs(A["impuls"][oa] - B["impuls"][ob]).max() < 1e-9 * np.abs(B["impuls"]).max(), s


CPU_MPI = os.path.join(ROOT, "oracle", "_ref", "imd_mpi_eam_nve_nbl")
GPU_MPI = os.path.join(ROOT, "oracle", "_ref", "imd_mpi_eam_nve_nbl_gpu")


def _mpiexec():
    import shutil
    for c in (shutil.which("mpiexec"), "/opt/conda/bin/mpiexec"):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.skipif(not (os.path.exists(CPU_MPI) and os.path.exists(GPU_MPI) and _mpiexec()), reason="MPI binaries / mpiexec not there")
@pytest.mark.parametrize("cpu_dim", ["2 1 1", "2 2 1"])
def test_reference_mpi_program_on_gpu_forces(cpu_dim):
    """The reference's MPI build on 2 and 4 ranks (mpiexec; the ranks share cuda:0) with the shim in place of
    imd_forces_nbl.o: the host keeps its decomposition, migration (fix_cells) and halo exchange (send_cells for the
    buffer cells and for EAM_DF), every rank's forces come from the device.  Per rank the same atoms with the same
    forces / energies / positions as the CPU build on the same decomposition over 30 steps."""
    nranks = int(np.prod([int(v) for v in cpu_dim.split()]))
    # msgbuf_size: the reference's own estimate of its MPI buffers is too small for this box (pack_cell aborts), CPU build too
    param = PARAM.replace("cpu_dim 1 1 1", "cpu_dim " + cpu_dim) + "msgbuf_size 4000000\n"
    out = {}
    for tag, exe in (("cpu", CPU_MPI), ("gpu", GPU_MPI)):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            open(os.path.join(d, "t.param"), "w").write(param)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0,1,30", IMD_SHIM_DEVICE="0")
            r = subprocess.run([_mpiexec(), "-n", str(nranks), exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-3000:]
            out[tag] = {(rk, s): po.read_probe(os.path.join(d, "probe.%d.%d.bin" % (rk, s))) for rk in range(nranks) for s in (0, 1, 30)}
            out[tag]["eng"] = np.loadtxt(os.path.join(d, "out.eng"))
    fmax = max(np.abs(out["cpu"][(rk, 30)]["kraft"]).max() for rk in range(nranks))
    for rk in range(nranks):
        for s in (0, 1, 30):
            a, b = out["gpu"][(rk, s)], out["cpu"][(rk, s)]
            ia, ib = _inner_ids(a), _inner_ids(b)
            oa, ob = np.argsort(ia), np.argsort(ib)
            assert np.array_equal(ia[oa], ib[ob]), "rank %d step %d: different atoms" % (rk, s)
            assert np.abs(a["kraft"][oa] - b["kraft"][ob]).max() < 1e-9 * fmax, (rk, s)
            assert np.abs(a["pot_eng"][oa] - b["pot_eng"][ob]).max() < 1e-10
            assert np.abs(a["eam_rho"][oa] - b["eam_rho"][ob]).max() < 1e-10
            assert abs(a["tot_pot_energy"] - b["tot_pot_energy"]) < 1e-9 * abs(b["tot_pot_energy"])
            assert np.abs(_inner_pos(a)[oa] - _inner_pos(b)[ob]).max() < 1e-9
        assert out["gpu"][(rk, 30)]["nbl_count"] == out["cpu"][(rk, 30)]["nbl_count"] >= 2
    assert np.allclose(out["gpu"]["eng"], out["cpu"]["eng"], rtol=1e-6, atol=1e-9)


@pytest.mark.skipif(not (os.path.exists(CPU_TTM) and os.path.exists(GPU_TTM) and _mpiexec()), reason="MPI binaries / mpiexec not there")
@pytest.mark.parametrize("cpu_dim", ["2 1 1", "2 2 1"])
def test_reference_mpi_ttm_program_on_gpu(cpu_dim):
    """The 3D-TTM program on 2 and 4 ranks with calc_forces AND calc_ttm from the shim: forces with the host's own halo
    exchange, the electron lattice as one block per rank (like l1/l2) with the per-sub-step ghost faces going through the
    library's shared-memory mailbox instead of ttm_fill_ghost_layers' MPI_Sendrecv.  Per rank: lattice block (what
    move_atoms_ttm reads, and temp, U, fd_k, ...) within 1e-9 of the CPU build on the same decomposition, atoms within 1e-9."""
    nranks = int(np.prod([int(v) for v in cpu_dim.split()]))
    param = PARAM_TTM.replace("cpu_dim 1 1 1", "cpu_dim " + cpu_dim)
    out = {}
    steps = (0, 1, 6, 12)
    for tag, exe in (("cpu", CPU_TTM), ("gpu", GPU_TTM)):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            _write_planar_eos(d)
            open(os.path.join(d, "t.param"), "w").write(param)
            env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS=",".join(map(str, steps)), IMD_SHIM_DEVICE="0")
            r = subprocess.run([_mpiexec(), "-n", str(nranks), exe, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-3000:]
            out[tag] = {(rk, s): po.read_probe(os.path.join(d, "probe.%d.%d.bin" % (rk, s))) for rk in range(nranks) for s in steps}
            out[tag]["lat"] = {(rk, s): po.read_ttm_probe(os.path.join(d, "ttm.%d.%d.bin" % (rk, s)))["stages"]["cfl"]
                               for rk in range(nranks) for s in steps}
    inner = (slice(1, -1),) * 3
    for rk in range(nranks):
        for s in steps:
            a, b = out["gpu"]["lat"][(rk, s)], out["cpu"]["lat"][(rk, s)]
            assert np.array_equal(a["lat"]["natoms"][inner], b["lat"]["natoms"][inner]), (rk, s)
            for f in ("temp", "temp2", "U", "U2", "xi", "md_temp", "dens", "vcom", "fd_k", "fd_g", "Ce"):
                x, y = a["lat"][f][inner], b["lat"][f][inner]
                assert np.abs(x - y).max() <= 1e-9 * np.abs(y).max(), (rk, s, f)
            assert abs(a["max_dt_ttm"] - b["max_dt_ttm"]) <= 1e-9 * abs(b["max_dt_ttm"])
            A, B = out["gpu"][(rk, s)], out["cpu"][(rk, s)]
            ia, ib = _inner_ids(A), _inner_ids(B)
            oa, ob = np.argsort(ia), np.argsort(ib)
            assert np.array_equal(ia[oa], ib[ob])
            assert np.abs(_inner_pos(A)[oa] - _inner_pos(B)[ob]).max() < 1e-9, (rk, s)
            assert np.array_equal(A["numneighs"][oa], B["numneighs"][ob])


GPU_TTM1D = os.path.join(ROOT, "oracle", "_ref", "imd_mpi_eam_ttm1d_tmm_ep_nbl_gpu")


@pytest.mark.skipif(not os.path.exists(GPU_TTM1D), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("level", [1, 2])
def test_reference_1d_program_with_its_laser_source_on_the_device(level):
    """The configuration the reference's Makefile builds -- imd_ttm.c's 1D model (-DTTM1D) with the transfer-matrix laser source
    of imd_tmm.c, here also with -DELECPRESS -- as a drop-in: the reference's main loop, init_ttm / tmm_init (they read the
    tables), its integrator (level 1) or the device's (level 2), with calc_forces and calc_ttm on the GPU.  The CPU side is
    tests/golden/ttm1d_ep_slab.npz: the same input run by the unmodified program on TWO ranks (it cannot run on one:
    ttm_fill_ghost_layers); IMD_SHIM_REF_RANKS=2 reproduces what the cut into two ranks does to the virtual lattice's
    neighbours.  Atoms of step 8 away from that run's rank boundary (the pressure force there depends on which rank owned an
    atom at the last re-binning)."""
    from imd_amd import tmm as tmm_mod, ttm as ttm_mod
    from tests import golden_util as gu
    from tests.golden import make_golden as mg
    from imd_amd import config_io
    G = gu.load("ttm1d_ep_slab")
    # the two-rank run drew its initial velocities per rank: the one-rank program starts from that run's atoms of step 0
    # (a binary checkpoint in the reference's own format) instead of generating its own
    par = str(G["param_file"]).replace("cpu_dim 2 1 1", "cpu_dim 1 1 1").replace("coordname _fcc", "coordname in.chkpt\nbox_from_header 1")
    par = par.replace("shiftx_front 8.0", "shiftx_front 0.0").replace("shiftx_rear 8.0", "shiftx_rear 0.0")
    assert "cpu_dim 1 1 1" in par and "in.chkpt" in par
    with tempfile.TemporaryDirectory() as top:
        d = os.path.join(top, "run"); os.mkdir(d)
        potential.write_synthetic_eam_files(d)
        config_io.write_config(os.path.join(d, "in.chkpt"), G["s0_nummer"], G["s0_sorte"], G["s0_masse"], G["s0_ort"], G["s0_impuls"],
                               np.zeros(len(G["s0_nummer"])), G["box"], (0, 1, 1), binary=True)
        eos = mg.write_planar_eos(top)
        ttm_mod.write_eos_table(os.path.join(top, "EOS_phase_from_r_ti.txt"), mg.tmm_phase_table(eos))
        ttm_mod.write_eos_table(os.path.join(top, "EOS_pe_from_r_te.txt"), mg.epress_table(eos))
        tmm_mod.write_arr(os.path.join(top, "alu_eps_bb.dat"), tmm_mod.synthetic_eps_bb())
        tmm_mod.write_arr(os.path.join(top, "K12.dat"), tmm_mod.synthetic_K12())
        open(os.path.join(d, "t.param"), "w").write(par)
        env = dict(os.environ, IMD_PROBE_DIR=d, IMD_PROBE_STEPS="0,4,8", IMD_SHIM_REF_RANKS="2", IMD_SHIM_LEVEL=str(level),
                   IMD_SHIM_SYNC_STEPS="0,4,8")
        r = subprocess.run([GPU_TTM1D, "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stdout[-3000:]
        P = {s: po.read_probe(os.path.join(d, "probe.0.%d.bin" % s)) for s in (0, 4, 8)}
    box = np.diag(G["box"])
    for s, tol in ((0, 1e-12), (4, 1e-8), (8, 1e-7)):
        ids, pos = _inner_ids(P[s]), _inner_pos(P[s])
        a, b = np.argsort(ids), np.argsort(G["s%d_nummer" % s])
        assert np.array_equal(ids[a], G["s%d_nummer" % s][b])
        ref = G["s%d_ort" % s][b]
        far = np.abs(ref[:, 0] - box[0] / 2) > 1.0
        dx = pos[a] - ref; dx[:, 1:] -= np.round(dx[:, 1:] / box[1:]) * box[1:]
        assert np.abs(dx[far]).max() < tol, (level, s, np.abs(dx[far]).max())
    # the laser and the pressure did act: the same atoms without them are elsewhere (the fixture of the run without the flag)
    T = gu.load("ttm1d_tmm_slab")
    a, b = np.argsort(_inner_ids(P[8])), np.argsort(T["s8_nummer"])
    assert np.abs(_inner_pos(P[8])[a] - T["s8_ort"][b]).max() > 1e-4
