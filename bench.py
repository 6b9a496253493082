#!/usr/bin/env python3
"""bench.py -- MD-steps/s of the device-resident force(+TTM) loop on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`
prints ONE JSON line on rank 0.  A "step" is one full MD step of the hot path
(calc_forces [+ calc_ttm] + move_atoms + check_nblist, list rebuilds included)
on a synthetic FCC-Al block with the synthetic tabulated EAM of SURVEY.md
Appendix B, state resident in HBM.  N>1: one rank per GPU, each rank owns an
equal block of atoms (weak scaling), launched by torch.distributed.run.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--box", type=int, nargs=3, default=[136, 136, 136],
                    help="FCC unit cells per direction PER GPU (136^3*4 = 10 061 824 atoms)")
    ap.add_argument("--temp", type=float, default=0.0258)
    ap.add_argument("--no-ttm", action="store_true")
    ap.add_argument("--fd-ext", type=int, nargs=3, default=[2, 2, 2],
                    help="MD cells per FD cell (BASELINE config 3: 2 2 2); 1 1 1 gives the 8x lattice an 8-GPU run replicates")
    ap.add_argument("--no-adv", action="store_true",
                    help="leave do_ADV out of calc_ttm (the reference compiles ADVMODE 1 and calls do_ADV(1.0) in every step,\n"
                         "imd_ttm_3D.c:17,58-60: on by default here as well)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="experiment: no HIP events around the kernel classes in the timed region (roofline then reports 0)")
    ap.add_argument("--laser", nargs="?", const="beer", default=None, choices=["beer", "fdtd"],
                    help="config 5: `--laser` / `--laser beer` = Beer-Lambert source laser_rescale_ttm (imd_laser.c:715) in every\n"
                         "step; `--laser fdtd` = the 1D Maxwell solver (imd_fdtd.c) on a slab: open x with vacuum in front of and\n"
                         "behind the atoms, ONE row of FD cells along x (fd_ext 1 ny nz), pulse centred on the timed region")
    ap.add_argument("--dl-tables", action="store_true",
                    help="with --laser fdtd: the cells' Drude-Lorentz parameters from fitDL's tricubic tables in every FILLMESH\n"
                         "(imd_ttm_3D.c:600,2919-2968; synthetic tables, DL1..5.txt are not part of the reference tree) instead of constants")
    ap.add_argument("--tables", choices=["shared", "distinct"], default="shared",
                    help="one atom type: `shared` = phi and rho tabulated on ONE r^2 grid (default; files written by one tool);\n"
                         "`distinct` = phi, rho and F on three different grids (begin, end, step, rows: potential.EAM_GRIDS) -- the\n"
                         "general pot_table_t of the reference (types.h:537-549), pass 1 in its distinct-grids form")
    ap.add_argument("--types", type=int, choices=[1, 2], default=1,
                    help="2 = two atom types on the Cu3Au (L1_2) decoration of the FCC lattice: 4-column pair / density tables of\n"
                         "3000-4000 rows with per-column begin / end / step (potential.write_synthetic_eam_files_2types), 2-column F")
    ap.add_argument("--types2-rows", type=float, default=1.0,
                    help="with --types 2: scale of the tables' row counts (1.0 = 3000-4000 rows per column: beyond the LDS, record kernels;\n"
                         "0.3 = 900-1200 rows: all of them sit in the LDS beside a tile)")
    ap.add_argument("--rec-lds", type=int, default=-1,
                    help="with --types 2: which tables the record kernels keep in the LDS (option tile_rec_lds; 1 pair_pot and 2 rho_h_tab\n"
                         "in pass 1, 4 rho_h_tab in pass 3; tables that cannot fit are dropped); default: the library's choice")
    ap.add_argument("--slab", action="store_true",
                    help="config 4's geometry: open x with 12 MD cells of vacuum in front of and behind the atoms (pbc 0 1 1)")
    ap.add_argument("--cpu-box", type=int, default=48, help="unit cells per direction of the CPU sample")
    ap.add_argument("--cpu-steps", type=int, default=30)
    ap.add_argument("--cpu-steps-mpi", type=int, default=40, help="steps of the multi-core (MPI + TTM) reference sample")
    ap.add_argument("--cpu-cells-per-rank", type=int, default=40, help="FCC unit cells per direction PER RANK of the MPI samples\n"
                    "(40: 256 000 atoms per rank -- sub-boxes comparable with config 3's 250 000 per core at 40 cores)")
    ap.add_argument("--cpu-ranks", type=int, default=16, help="MPI ranks of the 16-core CPU sample (the host-core share of a one-GPU job)")
    ap.add_argument("--cpu-ranks-all", type=int, default=0,
                    help="MPI ranks of the all-core CPU sample (0 = every core the affinity mask and the cgroup quota grant)")
    ap.add_argument("--cpu-all-timeout", type=float, default=240.0, help="give the all-core CPU sample up after this many seconds")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal: host-staged halos)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks on cuda:0")
    ap.add_argument("--ttm-lattice", choices=["auto", "decomposed", "replicated"], default=None,
                    help="several ranks: `decomposed` = one TTM lattice block per rank, the faces as ghost records written into the\n"
                         "neighbours' device memory from inside the sub-step kernel (host mailbox if that cannot be set up);\n"
                         "`replicated` = every rank diffuses the whole lattice, one all-gather per step; `auto` (default with\n"
                         "more than one rank) = both are built and timed for a few steps before the warm-up, the faster one that\n"
                         "works on every rank runs the benchmark (DESIGN.md 5).  One rank: replicated = the plain lattice.")
    ap.add_argument("--ttm-decomposed", action="store_true", help="= --ttm-lattice decomposed")
    ap.add_argument("--init-dist", action="store_true",
                    help="rehearsal with --force-distributed: a ONE-rank torch.distributed process group on --backend, so that the\n"
                         "host layer's nccl code path (object broadcast of the RCCL ids, all-reduces on device tensors) executes")
    ap.add_argument("--halo-overlap", type=int, default=0,
                    help="several ranks (or --force-distributed): pass 1 of the next step starts on interior tiles while its\n"
                         "position halo travels on a stream of its own; the value = CUs the first launch leaves to the transport")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="watchdog: seconds a rank (and the parent of self-spawned ranks) gives the whole run before it writes its\n"
                         "status, exits non-zero and the parent prints every rank's last status; 0 = off")
    ap.add_argument("--status-dir", default=None, help=argparse.SUPPRESS)      # where the ranks leave their status files (set by the parent)
    ap.add_argument("--hook-rank-hang", type=int, default=-1, help=argparse.SUPPRESS)     # test hook: this rank sleeps in the warm-up
    ap.add_argument("--hook-remote-fail", type=int, default=0, help=argparse.SUPPRESS)     # test hook: this rank (1-based) cannot set its ghost records up
    ap.add_argument("--allow-experiments", action="store_true",
                    help="run although the library was built with experiment macros (imdgpu_get_info(\"experiments\") != 0): the line\n"
                         "then carries \"experiments\": mask and is not a product measurement")
    ap.add_argument("--force-distributed", action="store_true",
                    help="rehearsal: run the multi-rank host layer (halo plan, gather/scatter, fork/join) on one rank")
    return ap.parse_args()


TRIAL_WARM, TRIAL_STEPS = 6, 24          # `--ttm-lattice auto`: steps of each candidate before / inside its timing


def build_system(args, rank, world, device, decomposed=False):
    """One rank's share: its own block of args.box FCC unit cells (weak scaling); the ranks form the most
    cubic grid (8 -> 2x2x2: six faces on six different xGMI links)."""
    from imd_amd import lattice, potential
    from imd_amd.system import GpuSystem, global_cell_dim
    a = 4.05
    if args.types == 2:
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files_2types(d, rows=args.types2_rows)
            pair, c1 = potential.read_pot_table(os.path.join(d, "phi.pt"), 4, True)
            rho, c2 = potential.read_pot_table(os.path.join(d, "rho.pt"), 4, True)
            emb, _ = potential.read_pot_table(os.path.join(d, "F.pt"), 2, False)
        r2cut = max(c1, c2)
    else:
        pair, rho, emb, r2cut = potential.synthetic_eam_tables(grids=args.tables)
    nbl_margin = 0.5
    cellsz = (np.sqrt(r2cut) + nbl_margin) ** 2
    ttm = not args.no_ttm
    fd_ext = tuple(args.fd_ext)
    if world == 1:
        cpu_dim, coord = (1, 1, 1), (0, 0, 0)
    else:
        from imd_amd.parallel import choose_cpu_dim, rank_to_coord
        cpu_dim = choose_cpu_dim(world)
        coord = rank_to_coord(rank, cpu_dim)
    gbox_param = [args.box[d] * cpu_dim[d] for d in range(3)]
    box = lattice.fcc_box(gbox_param, a)
    pbc, shift_x = (1, 1, 1), 0.0
    if args.laser == "fdtd":
        if world > 1:
            raise SystemExit("bench.py --laser fdtd: one GPU (the reference's 1D solver row is not decomposed either)")
        # slab: 45 MD cells of vacuum in front (PML 15 + source + run-up), 25 behind; the lattice is one row of FD cells
        gd0 = global_cell_dim(box, cellsz)
        hx = box[0, 0] / gd0[0]
        shift_x = 45 * hx
        box = box.copy(); box[0, 0] += 70 * hx
        pbc = (0, 1, 1)
        gdim = (gd0[0] + 70, gd0[1], gd0[2])
        fd_ext = (1, gd0[1], gd0[2])
    else:
        cd = tuple(cpu_dim[d] * (fd_ext[d] if ttm else 1) for d in range(3))
        if args.slab:                 # vacuum on both x sides (shiftx_front / shiftx_rear of the reference's generator)
            pbc = (0, 1, 1)
            gd0 = global_cell_dim(box, cellsz, cd)
            hx = box[0, 0] / gd0[0]
            nvac = ((12 + cd[0] - 1) // cd[0]) * cd[0]            # a multiple of cpu_dim.x * fd_ext.x cells per side
            shift_x = nvac * hx
            box = box.copy(); box[0, 0] += 2 * nvac * hx
            gdim = (gd0[0] + 2 * nvac, gd0[1], gd0[2])
        else:
            gdim = global_cell_dim(box, cellsz, cd)
    cell_dim = tuple(gdim[d] // cpu_dim[d] + 2 for d in range(3))
    sysm = GpuSystem(box, pbc, cell_dim, args.types, cellsz, nbl_margin, 0.1, device=device, global_cell_dim=gdim,
                     cpu_dim=cpu_dim, my_coord=coord, count_ttm_neighbours=ttm,
                     external_halo=args.force_distributed,
                     ttm_decomposed=ttm and (world > 1 or args.force_distributed) and decomposed)
    sysm.upload_tables(pair, rho, emb)
    exp = sysm.get_info("experiments")
    if exp and not args.allow_experiments:
        raise SystemExit("bench.py: libimdgpu.so was built with experiment macros (mask %d: see imdgpu_experiment_mask); "
                         "rebuild without EXTRA flags or pass --allow-experiments" % exp)
    if args.hook_remote_fail:
        sysm.set_option("ttm_remote_fail", args.hook_remote_fail)
    if args.rec_lds >= 0:
        sysm.set_option("tile_rec_lds", args.rec_lds)
    if args.halo_overlap > 0:
        sysm.set_option("halo_overlap", args.halo_overlap)
    if ttm and adv_on(args):
        sysm.set_option("ttm_adv", 1)            # calc_ttm's own sequence: update_fd, do_ADV(1.0), activation, ... (:57-60)
    if ttm:
        from imd_amd import ttm as ttm_mod
        ttm_mod.attach_default(sysm, fd_ext=fd_ext, diff_substeps=100, fd_min_atoms=5)
        if args.laser == "fdtd":
            # 800 nm, 2e11 W/cm^2, sigma_t 60 fs centred 150 MD steps in (the timed region follows 50 warm-up steps): the
            # solver is active during the whole timed region; source 30 cells in front of the surface
            sysm.init_fdtd(I0=2e15, lam=800e-9, sigma_t=60e-15, t_0=150 * 0.1 * 10.18e-15, srcx=25 * (box[0, 0] / gdim[0]) * 1e-10,
                           bw=15, Sc=1.0)
            if args.dl_tables:
                from imd_amd import tricub
                sysm.fdtd_set_dl_tables(tricub.synthetic_dl_tables(500.0, 6000.0))
        elif args.laser:          # a gentle pulse along x peaking at step 150: mu = 1/(20 A), 0.4 eV/A^2 fluence, sigma_t = 5 time units
            sig_t, mu, sig_e = 5.0, 0.05, 0.4
            sysm.set_laser(direction=(1, 0, 0), offset=0.0, mu=mu, p_peak=mu * sig_e / sig_t / np.sqrt(2 * np.pi),
                           t_0=15.0, sigma_t_squared=sig_t * sig_t)
    pos = lattice.fcc_positions(args.box, a) + np.array([coord[d] * args.box[d] * a for d in range(3)]) + np.array([shift_x, 0.0, 0.0])
    ident = np.arange(len(pos), dtype=np.int64) + rank * len(pos)
    if args.types == 2:
        # L1_2: the cube corners of every FCC unit cell are type 1 (a quarter of the atoms), the face centres type 0
        half = np.rint((pos - np.array([shift_x, 0.0, 0.0])) / (a / 2.0) - 0.5).astype(np.int64)
        sorte = np.all(half % 2 == 0, axis=1).astype(np.int32)
        masse = np.where(sorte == 1, 63.546, 26.9815)
        mom = lattice.maxwell_momenta(len(pos), 1.0, args.temp, seed=12345 + rank) * np.sqrt(masse)[:, None]
        mom -= mom.mean(axis=0, keepdims=True)
        sysm.upload_atoms(pos, mom, masse, sorte, (ident % (2 ** 31 - 1)).astype(np.int32))
    else:
        mom = lattice.maxwell_momenta(len(pos), 26.9815, args.temp, seed=12345 + rank)
        sysm.upload_atoms(pos, mom, 26.9815, None, (ident % (2 ** 31 - 1)).astype(np.int32))
    md = None
    if world > 1 or args.force_distributed:
        from imd_amd.parallel import Decomposition, DistributedMD, GpuOps
        dec = Decomposition(rank, cpu_dim, gdim, box, pbc)
        md = DistributedMD(dec, GpuOps(sysm, device), ttm=ttm)
    return sysm, md, len(pos), gdim, cpu_dim


def adv_on(args):
    """do_ADV belongs to the reference's calc_ttm (ADVMODE 1); not with the one-row lattice of the Maxwell-solver slab, whose
    FD cells span thousands of MD cells (no flux counters there)"""
    return not args.no_ttm and not args.no_adv and args.laser != "fdtd"


def _cpu_env():
    """environment of the CPU reference runs: no profiler preloads (under rocprofv3 every child would otherwise load the
    tool library and touch the GPU), probe dumps off"""
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTRACER", "ROCTX"))}
    env["IMD_PROBE_STEPS"] = "-1"
    return env


def _host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _find_mpiexec():
    import shutil
    for c in (shutil.which("mpiexec"), "/opt/conda/bin/mpiexec"):
        if c and os.path.exists(c):
            return c
    return None


def _cpu_quota():
    """cores a cgroup CPU quota leaves to this job, or None (cgroup v2 cpu.max, v1 cfs_quota_us)"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(float(q) / float(per)))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except Exception:
        pass
    return None


def _rank_grid(n):
    """cpu_dim for n MPI ranks: powers of two split as evenly as possible, x first (x >= 2 so that do_ADV's exchange with the
    x neighbour exists: with cpu_dim.x = 1 the reference's do_COMMFLUX cannot run)"""
    n = max(1, int(n))
    p = 1
    while p * 2 <= n:
        p *= 2
    d = [1, 1, 1]
    k = 0
    while p > 1:
        d[k % 3] *= 2; p //= 2; k += 1
    return tuple(d)


def cpu_baseline_mpi_ttm(args, nranks_want, per_rank_cells, steps, timeout):
    """The workload of the GPU number on the host cores: the reference's own MPI build with the 3D TTM
    (oracle/_ref/imd_mpi_eam_ttm3d_nbl: EAM forces through imd_forces_nbl.c, calc_ttm's first block -- update_fd, do_ADV,
    activation, FILLMESH, 100 x (do_DIFF, FILLMESH, ghosts), CFL --, move_atoms_ttm, list maintenance, halo exchange), one
    rank per core, on a bounded sample of per_rank_cells^3 FCC unit cells per rank (sub-boxes like config 3's)."""
    from imd_amd import potential, ttm as ttm_mod
    exe = os.path.join(ROOT, "oracle", "_ref", "imd_mpi_eam_ttm3d_nbl")
    mpiexec = _find_mpiexec()
    if args.no_ttm or not os.path.exists(exe) or mpiexec is None or args.types != 1:
        return None
    cpu_dim = _rank_grid(nranks_want)
    nranks = cpu_dim[0] * cpu_dim[1] * cpu_dim[2]
    nb = [per_rank_cells * c for c in cpu_dim]
    natoms = 4 * nb[0] * nb[1] * nb[2]
    adv = adv_on(args) and cpu_dim[0] >= 2
    with tempfile.TemporaryDirectory() as d:
        potential.write_synthetic_eam_files(d, grids=args.tables)
        cve, ee = ttm_mod.synthetic_eos()
        ttm_mod.write_eos_table(os.path.join(d, "EOS_cve_from_r_te.txt"), cve)
        ttm_mod.write_eos_table(os.path.join(d, "EOS_ee_from_r_tesqrt.txt"), ee)
        open(os.path.join(d, "b.param"), "w").write(
            "coordname _fcc\noutfiles out\nensemble ttm\nmaxsteps %d\ntimestep 0.1\nntypes 1\ntotal_types 1\n"
            "masses 26.9815\ntypes 0\nbox_param %d %d %d\nbox_unit 4.05\npbc_dirs 1 1 1\nstarttemp %g\nseed 12345\n"
            "core_potential_file phi.pt\nembedding_energy_file F.pt\natomic_e-density_file rho.pt\n"
            "eng_int 100000\ncheckpt_int 100000\nnbl_margin 0.5\nnbl_size 1.5\ncpu_dim %d %d %d\n"
            "fd_ext %d %d %d\nfd_n_timesteps 100\nfd_min_atoms 5\natomic_weight 26.9815\natomic_charge 13\n"
            "ttm_int 100000\nI0 1.0\nlambda 800e-9\nlaser_sigma_t 100e-15\nlaser_t_0 1.0\n"
            % ((steps, nb[0], nb[1], nb[2], args.temp) + cpu_dim + tuple(args.fd_ext)))
        env = _cpu_env()
        if adv:
            env["IMD_PROBE_ADV"] = "1"           # the probe-driven calc_ttm then calls the reference's do_ADV(1.0) (ref_probe_ttm.c)
        t0 = time.time()
        try:
            r = subprocess.run([mpiexec, "-n", str(nranks), exe, "-p", "b.param"], cwd=d, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=timeout)
        except subprocess.TimeoutExpired:
            return {"value": None, "unit": "atom-steps/s", "cores": nranks, "cores_used": nranks, "kind": "reference",
                    "sample": "reference binary imd_mpi_eam_ttm3d_nbl on %d ranks, %d atoms x %d steps: gave up after %.0f s"
                              % (nranks, natoms, steps + 1, timeout)}
        except OSError:
            return None
        wall = time.time() - t0
    if r.returncode != 0:
        return None
    secs = None
    for line in r.stdout.splitlines():
        if "seconds excluding setup time" in line:
            secs = float(line.split()[0])                 # imd.c prints cputime / num_cpus here = seconds of the main loop
    if not secs:
        return None
    return {"value": natoms * (steps + 1) / secs, "unit": "atom-steps/s", "cores": nranks, "cores_used": nranks,
            "kind": "reference",
            "sample": "reference binary imd_mpi_eam_ttm3d_nbl (gcc -O2, MPI, cpu_dim %d %d %d = %d ranks), %d atoms (%d per rank) "
                      "x %d steps, EAM + 3D TTM (fd_ext %d %d %d, 100 sub-steps, do_ADV %s, no laser, calc_ttm's first block), main "
                      "loop incl. list rebuilds, %.1f s main loop, %.1f s wall"
                      % (cpu_dim + (nranks, natoms, natoms // nranks, steps + 1) + tuple(args.fd_ext) + ("on" if adv else "off", secs, wall))}


def cpu_baseline(args):
    """Primary: the reference's MPI + TTM build on EVERY core this job may use (affinity mask, capped by a cgroup CPU quota
    if there is one) -- BASELINE.md's "all physical host cores".  Beside it: the same build on 16 ranks (the host-core share
    of a one-GPU job) and the serial NVE reference binary on one core; they are also the fallbacks, in that order."""
    ser = cpu_baseline_serial(args)
    avail, quota = _host_cores(), _cpu_quota()
    grant = min(avail, quota) if quota else avail
    r16 = cpu_baseline_mpi_ttm(args, min(args.cpu_ranks, grant), args.cpu_cells_per_rank, args.cpu_steps_mpi, 400)
    want_all = args.cpu_ranks_all if args.cpu_ranks_all > 0 else grant
    rall = None
    if r16 is not None and r16.get("value") and _rank_grid(want_all) != _rank_grid(min(args.cpu_ranks, grant)):
        # 40^3 unit cells = 256 000 atoms per rank; steps so that the main loop is ~10-20 s at the 16-rank per-core rate
        per_core = r16["value"] / r16["cores_used"]
        n_all = int(np.prod(_rank_grid(want_all)))
        steps = int(min(40, max(8, 15.0 * per_core / 256000.0)))
        rall = cpu_baseline_mpi_ttm(args, n_all, args.cpu_cells_per_rank, steps, args.cpu_all_timeout)
    par = rall if (rall is not None and rall.get("value")) else r16
    if par is None:
        return ser
    par = dict(par)
    par["cores_available"] = avail
    par["cores_cgroup_quota"] = quota
    if r16 is not None:
        par["ranks16"] = {k: r16[k] for k in ("value", "unit", "cores", "kind", "sample")}
    if rall is not None and not rall.get("value"):
        par["all_cores_attempt"] = rall["sample"]
    par["serial"] = {k: ser[k] for k in ("value", "unit", "cores", "kind", "sample")}
    par["note"] = ("cores = cores_used = MPI ranks started: every core the affinity mask (cores_available) and the cgroup quota "
                   "(cores_cgroup_quota, null = none) grant, rounded down to a power of two; ranks16 = the same build on the "
                   "host-core share of a one-GPU job; serial = the NVE binary on one core")
    return par


def cpu_baseline_serial(args):
    """The REAL reference binary (oracle/_ref, built here from /root/reference and
    shipped to the GPU box as a binary) on a bounded sample, 1 core; falls back to
    the oracle port when the binary is absent."""
    from imd_amd import potential
    exe = os.path.join(ROOT, "oracle", "_ref", "imd_eam_nve_nbl")
    nb, steps = args.cpu_box, args.cpu_steps
    natoms = 4 * nb ** 3
    if os.path.exists(exe):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d, grids=args.tables)
            open(os.path.join(d, "b.param"), "w").write(
                "coordname _fcc\noutfiles out\nensemble nve\nmaxsteps %d\ntimestep 0.1\nntypes 1\ntotal_types 1\n"
                "masses 26.9815\ntypes 0\nbox_param %d %d %d\nbox_unit 4.05\npbc_dirs 1 1 1\nstarttemp %g\nseed 12345\n"
                "core_potential_file phi.pt\nembedding_energy_file F.pt\natomic_e-density_file rho.pt\n"
                "eng_int 100000\ncheckpt_int 100000\nnbl_margin 0.5\nnbl_size 1.5\ncpu_dim 1 1 1\n"
                % (steps, nb, nb, nb, args.temp))
            env = _cpu_env()
            t0 = time.time()
            r = subprocess.run([exe, "-p", "b.param"], cwd=d, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            wall = time.time() - t0
            per = None
            for line in r.stdout.splitlines():
                if "cpuseconds per step and atom" in line:
                    per = float(line.split()[0])
            if r.returncode == 0 and per:
                return {"value": 1.0 / per, "unit": "atom-steps/s", "cores": 1, "cores_used": 1, "cores_available": _host_cores(),
                        "kind": "reference",
                        "sample": "reference binary imd_eam_nve_nbl (gcc -O2), %d atoms x %d steps, EAM NVE, "
                                  "main loop incl. list rebuilds, %.1f s wall" % (natoms, steps + 1, wall)}
    # port: the oracle restatement
    from oracle import pyoracle as po
    from imd_amd import lattice
    pos = lattice.fcc_positions((nb, nb, nb))
    box = lattice.fcc_box((nb, nb, nb))
    pair, rho, emb, r2cut = potential.synthetic_eam_tables(grids=args.tables)
    cellsz = (np.sqrt(r2cut) + 0.5) ** 2
    gd = po.global_cell_dim(box, cellsz)
    s = po.OracleSystem([g + 2 for g in gd], 48, 1, box, (1, 1, 1), cellsz, 0.5, 0.1)
    for w, t in enumerate((pair, rho, emb)):
        s.set_table(w, t.as_dict())
    mom = lattice.maxwell_momenta(len(pos), 26.9815, args.temp)
    s.fill_from_positions(pos, mom, 26.9815)
    t0 = time.time()
    for st in range(steps):
        s.calc_forces(st); s.move_atoms_nve(); s.check_nblist()
    wall = time.time() - t0
    return {"value": natoms * steps / wall, "unit": "atom-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle C restatement, %d atoms x %d steps, EAM NVE" % (natoms, steps)}


def measured_traffic(kernel, natoms, kernel_rev):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/profile_bench.sh:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload).  Counters cannot be read
    from inside the process, so the figure is looked up by (full kernel name incl. template arguments, workload size,
    kernel revision of the library); an entry measured on another build of the force kernels (`_kernel_rev` differs
    from imdgpu_get_info("kernel_rev")) is REFUSED -> traffic null."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("_atoms_per_gpu") != natoms or t.get("_kernel_rev") != kernel_rev:
            continue
        for k, v in t.items():
            if isinstance(v, dict) and k.replace("void ", "").startswith(kernel) and v.get("hbm_bytes"):
                return v["hbm_bytes"], os.path.relpath(f, ROOT)
    return None, None


def measured_units(kernel, natoms, kernel_rev):
    """What bounds the kernel besides HBM: busy fractions of the LDS array and of the vector ALUs during its launches, from the
    committed SQ counter passes (tools/pmc_sq.sh -> profiles/*sq_counters*.json: lds_busy = SQ_LDS_IDX_ACTIVE / (CUs x cycles),
    valu_busy = 4 x SQ_ACTIVE_INST_VALU / (SIMDs x cycles), cycles = GRBM_GUI_ACTIVE / 8 of the same launches).  Looked up like the
    traffic: full kernel name, atoms per GPU and the library's kernel revision must match, else null."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*sq_counters*.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("_atoms_per_gpu") != natoms or t.get("_kernel_rev") != kernel_rev:
            continue
        for k, v in t.items():
            if isinstance(v, dict) and k.replace("void ", "").startswith(kernel) and "valu_busy" in v and "lds_busy" in v:
                return {"lds": v["lds_busy"], "valu": v["valu_busy"], "lds_bank_conflict_share": v.get("lds_conflict_share"),
                        "clock_ghz": v.get("clock_ghz_a"), "avg_launch_ms_under_counters": v.get("avg_ns_a", 0) / 1e6,
                        "source": os.path.relpath(f, ROOT)}
    return None


def _rank_status(args, rank, text):
    """a rank's last known phase, one small file per rank (the parent prints them when the run times out or fails)"""
    if args.status_dir:
        try:
            with open(os.path.join(args.status_dir, "rank%d.status" % rank), "w") as f:
                f.write("%s at +%.1f s (pid %d)\n" % (text, time.time() - _T0, os.getpid()))
        except OSError:
            pass


_T0 = time.time()


def _arm_watchdog(args, rank):
    """A rank that is still running after --rank-timeout seconds says where it is and ends the process with exit code 3 -- no re-exec,
    no retry: torch.distributed.run then takes the other ranks down and the parent reports (a rank stuck in a collective or a device
    wait cannot be unwound from Python; ending the process is the only exit every such state has)."""
    if args.rank_timeout <= 0:
        return None
    import threading

    def fire():
        _rank_status(args, rank, "WATCHDOG: still running after %.0f s; last phase: %s" % (args.rank_timeout, _PHASE[0]))
        sys.stderr.write("bench.py: rank %d watchdog fired after %.0f s in phase '%s'\n" % (rank, args.rank_timeout, _PHASE[0]))
        sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(args.rank_timeout, fire)
    t.daemon = True
    t.start()
    return t


_PHASE = ["start"]


def _phase(args, rank, text):
    _PHASE[0] = text
    _rank_status(args, rank, text)


def _spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has not
    touched the GPU yet) through torch.distributed.run and relay rank 0's JSON line.  The parent is the outer watchdog: when the
    launcher has not returned --rank-timeout + 60 s after it started, its process group is killed; on any failure every rank's last
    status goes to stderr and the exit code is non-zero."""
    import signal
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    sdir = tempfile.mkdtemp(prefix="bench_status_")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + ["--status-dir", sdir]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
    timed_out = False
    try:
        out, _ = p.communicate(timeout=(args.rank_timeout + 60.0) if args.rank_timeout > 0 else None)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(p.pid, signal.SIGKILL)          # the launcher and its ranks: the process group this parent created, nothing else
        except OSError:
            pass
        out, _ = p.communicate()
    line = [l for l in (out or "").splitlines() if l.startswith("{") and '"metric"' in l]
    if line and p.returncode == 0 and not timed_out:
        print(line[-1])
    else:
        sys.stdout.write(out or "")
        sys.stderr.write("bench.py --gpus %d: %s (launcher exit code %s); last status of every rank:\n"
                         % (args.gpus, "timed out" if timed_out else "failed", p.returncode))
        for r in range(args.gpus):
            try:
                st = open(os.path.join(sdir, "rank%d.status" % r)).read().strip()
            except OSError:
                st = "no status file: the rank never got as far as parsing its arguments"
            sys.stderr.write("  rank %d: %s\n" % (r, st))
    import shutil
    shutil.rmtree(sdir, ignore_errors=True)
    raise SystemExit(0 if (line and p.returncode == 0 and not timed_out) else (p.returncode or 1))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_ranks(args)
    # stdout carries exactly ONE JSON line: everything native libraries print while they start up (RCCL's version banner
    # goes to stdout) is diverted to stderr; the real stdout comes back just before rank 0 prints the result
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _phase(args, rank, "started")
    _arm_watchdog(args, rank)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, "
                         "or without a launcher)" % (args.gpus, world, args.gpus))
    if args.share_gpu:
        local = 0
    if world > 1 or (args.force_distributed and args.init_dist):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:           # a one-rank rehearsal without a launcher: any free port
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    def barrier(s=None):
        (s or sysm).synchronize()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()

    def agree_max(v):
        if not dist.is_initialized():
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    mode = "decomposed" if args.ttm_decomposed else args.ttm_lattice
    if mode is None:
        mode = "auto" if (world > 1 and not args.no_ttm) else "replicated"
    step, lattice_trial = 0, None
    if mode == "auto" and world > 1 and not args.no_ttm:
        # both lattice modes for a few steps on the real workload; every rank takes part in both and in the verdict.  A mode
        # that fails anywhere (the library reports set-up and time-out failures on all ranks alike) is out.
        lattice_trial, cands = {}, {}
        for m in ("decomposed", "replicated"):
            ms, err, built = 1e30, None, None
            try:
                built = build_system(args, rank, world, local, decomposed=(m == "decomposed"))
                for st in range(TRIAL_WARM):
                    built[1].md_step(st)
                barrier(built[0]); t0 = time.perf_counter()
                for st in range(TRIAL_WARM, TRIAL_WARM + TRIAL_STEPS):
                    built[1].md_step(st)
                barrier(built[0])
                ms = (time.perf_counter() - t0) / TRIAL_STEPS * 1e3
            except Exception as e:          # noqa: BLE001 -- whatever it was, this mode is out
                err = "%s: %s" % (type(e).__name__, str(e)[:300])
            ms = agree_max(ms)
            if ms >= 1e29:
                if built is not None:
                    try:
                        built[0].close()
                    except Exception:       # noqa: BLE001
                        pass
                lattice_trial[m] = {"ok": False, "error": err or "failed on another rank"}
            else:
                remote = built[0].get_info("ttm_remote") if m == "decomposed" else 0
                ek = built[1].global_sums()["tot_kin_energy"]            # collective; the same steps from the same start in both modes
                lattice_trial[m] = {"ok": True, "ms_per_step": ms, "steps": TRIAL_STEPS, "tot_kin_energy": ek,
                                    "faces": ("ghost records in the neighbours' device memory (%s)" % ("fine-grained" if remote == 2 else "plain: ranks share a GPU")
                                              if remote else "host mailbox") if m == "decomposed" else "all-gather"}
                cands[m] = (ms, built)
        # the decomposed lattice's hand-offs across GPUs are coherent by construction (fine-grained memory, system scope) -- and
        # checked here where it counts: both modes ran the same steps from the same start, so the electron-phonon coupling must
        # have left the same kinetic energy; a decomposed lattice that disagrees with the replicated one is out
        if "decomposed" in cands and "replicated" in cands:
            a, b = lattice_trial["decomposed"]["tot_kin_energy"], lattice_trial["replicated"]["tot_kin_energy"]
            if not abs(a - b) <= 1e-7 * abs(b):
                lattice_trial["decomposed"].update(ok=False, error="kinetic energy after the trial differs from the replicated lattice's: %.12e vs %.12e" % (a, b))
                cands.pop("decomposed")[1][0].close()
        if not cands:
            raise SystemExit("bench.py: neither lattice mode ran: %s" % json.dumps(lattice_trial))
        mode = min(cands, key=lambda k: cands[k][0])
        for k, (_, b) in cands.items():
            if k != mode:
                b[0].close()
        sysm, md, natoms, gdim, cpu_dim = cands[mode][1]
        step = TRIAL_WARM + TRIAL_STEPS
    else:
        sysm, md, natoms, gdim, cpu_dim = build_system(args, rank, world, local, decomposed=(mode == "decomposed"))
    do_step = md.md_step if md is not None else sysm.md_step
    _phase(args, rank, "system built (%d atoms on %s), warm-up" % (natoms, sysm.device_identity()))
    if args.hook_rank_hang == rank:
        time.sleep(10 ** 6)
    while step < args.warmup:
        do_step(step); step += 1
    barrier()
    _phase(args, rank, "timed region")
    # who is who: every rank's GPU (PCI bus id, UUID) and what RCCL itself reports for the communicators the library created
    ident = "rank %d: %s" % (rank, sysm.device_identity())
    rccl_here = {"rccl_ranks": sysm.get_info("rccl_ranks"), "rccl_comms": sysm.get_info("rccl_comms"), "rccl_device": sysm.get_info("rccl_device")}
    if dist.is_initialized():
        idents, rccls = [None] * world, [None] * world
        dist.all_gather_object(idents, ident); dist.all_gather_object(rccls, rccl_here)
    else:
        idents, rccls = [ident], [rccl_here]
    sysm.profile(not args.no_kernel_timing); sysm.profile_reset()
    nbl0 = sysm.get_sums()["nbl_count"]
    standin0 = sysm.get_info("p3_standin_passes")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        do_step(step); step += 1
    barrier()
    dt = time.perf_counter() - t0
    _phase(args, rank, "timed region done, collecting results")
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    sums = sysm.get_sums()
    rebuilds = sums["nbl_count"] - nbl0
    if md is not None:
        g = md.global_sums()
        sums["tot_pot_energy"], sums["tot_kin_energy"] = g["tot_pot_energy"] / world, g["tot_kin_energy"] / world
    entries, maxn = sysm.nblist_stats()
    M = entries / (2.0 * natoms)                       # half-list entries per atom
    prof = {k: sysm.profile_get(k) for k in ("pass1", "pass2", "pass3", "ghosts", "nblist", "fix_cells", "integrate",
                                             "reduce", "ttm_update_fd", "ttm_diff", "ttm_fillmesh", "pass1b")}
    overlap_passes = sysm.get_info("halo_overlap_passes")
    if prof["pass1b"][0]:          # option halo_overlap: pass 1 ran as two launches from one tile queue; one pass = both
        prof["pass1"] = (max(prof["pass1"][0], 1), prof["pass1"][1] + prof["pass1b"][1])
    sysm.profile(False)
    if rank == 0:
        # The two force passes share the top of the profile; `roofline` prices whichever of them took longer in THIS run
        # (the dominant kernel) and `roofline_other` the second.  Algorithmic bytes per atom: pass 1 = 72 + 4M,
        # plus pass 2's 32 (rho 8 + dF 8 + pot_eng RMW 16) when the embedding pass is folded into the same launch;
        # pass 3 = 84 + 4M.  Durations: HIP events on the library's stream around every launch of the timed region.
        fused = bool(sysm.get_info("fused_embedding"))
        ttmn = "true" if not args.no_ttm else "false"
        tiles = sysm.get_info("list_mode") == 1
        krev = sysm.get_info("kernel_rev")
        fcap = sysm.get_info("fast_cap")          # plane stride the neighbour loops are compiled for (template argument; 0: run time)
        p1form = sysm.get_info("pass1_form")      # 2: one type, shared grid (interleaved table); 1: one type, distinct grids; 0: general kernels
        nt1 = "true" if args.types == 1 else "false"
        rec_lds = sysm.get_info("rec_lds")        # several types: which tables the record kernels keep in the LDS (1 phi, 2 rho in pass 1, 4 rho in pass 3)

        def price(which):
            n, ms = prof[which]
            if which == "pass1":
                byts = natoms * (72.0 + 4.0 * M + (32.0 if fused else 0.0))
                if tiles and p1form == 3:
                    kn = "k_pass1_tile_rec<%s, %d, %s, %s>" % (ttmn, fcap, "true" if rec_lds & 1 else "false", "true" if rec_lds & 2 else "false")
                elif tiles and p1form:
                    kn = "k_pass1_tile_merged<%s, %s, %d, false, %s>" % (ttmn, "true" if fused else "false", fcap, "true" if p1form == 1 else "false")
                elif tiles:
                    kn = "k_pass1_tile<true, true, %s, %s>" % (ttmn, nt1)
                else:
                    kn = ("k_pass1_gt<true, true, %s>" % ttmn) if sysm.get_info("global_tables_in_place") else ("k_pass1<true, true, %s, %s>" % (ttmn, nt1))
                label = kn + (" (pass 1: pair + rho, embedding pass folded in)" if fused else " (pass 1: pair + rho)")
                if p1form == 1:
                    label += " [phi and rho on different grids]"
                if p1form == 3:
                    label += " [several atom types; in the LDS: %s; through L2 as coefficient records: %s]" % (
                        " + ".join(n for b, n in ((1, "pair_pot"), (2, "rho_h_tab")) if rec_lds & b) or "no table",
                        " + ".join(n for b, n in ((1, "pair_pot"), (2, "rho_h_tab")) if not rec_lds & b) or "no table")
                if prof["pass1b"][0]:
                    kn = kn.replace(", false, false>", ", true, false>")
                    label = kn + " (pass 1 from a tile queue under the position halo, two launches per pass: pair + rho, embedding pass folded in)"
            else:
                byts = natoms * (84.0 + 4.0 * M)
                # the coefficient form steps aside at run time when a pair falls below the rows it holds; the three-row kernel
                # then does that pass under the same timing class: name the coefficient kernel only if that never happened
                # in the timed region, else the stand-in, and say how often
                standin = sysm.get_info("p3_standin_passes") - standin0
                coef = tiles and sysm.get_info("p3_coef_rows") > 0 and standin == 0          # one atom type here
                kn = ("k_pass3_tile_rec%s<%d, %s>" % ("" if args.no_ttm else "_lean", fcap, "true" if rec_lds & 4 else "false")) if (tiles and p1form == 3) else ("k_pass3_tile_coef<4, %d>" % fcap) if coef else (("k_pass3_tile<%s, %d, %d>" % (nt1, sysm.get_info("pass3_ring"), fcap)) if tiles else ("k_pass3_gt" if sysm.get_info("global_tables_in_place") else "k_pass3<%s>" % nt1))
                label = kn + (" (pass 3: EAM force pass, gradient-coefficient table)" if coef else " (pass 3: EAM force pass)")
                if standin > 0:
                    label += " [stood in for the coefficient form in %d of %d passes]" % (standin, n)
            avg = ms / max(n, 1)
            a = byts / (avg * 1e-3) / 1e9 if n else 0.0
            tr, src = measured_traffic(kn, natoms, krev)
            un = measured_units(kn, natoms, krev)
            if un is not None:
                un = dict(un, hbm=a / HBM_PEAK_GBS,
                          note="fractions of each unit's own peak during the kernel's launches: lds / valu = busy cycles of the LDS array / "
                               "the vector ALUs (SQ counters of the same kernel revision), hbm = algorithmic bytes over the launch at 8 TB/s")
            return {"bound": "hbm", "unit_busy": un, "kernel": label, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                    "traffic": tr, "traffic_source": src, "algorithmic_bytes_per_launch": byts, "avg_launch_ms": avg,
                    "launches": n, "kernel_rev": krev}
        lattice_note = ""
        if md is not None and overlap_passes > 0:
            lattice_note += ", position halo under interior tiles of pass 1 (%d passes)" % overlap_passes
        if md is not None and not args.no_ttm:
            lattice_note += (", TTM lattice decomposed (one block per rank, faces: %s)" %
                            ("ghost records in the neighbours' device memory" if sysm.get_info("ttm_remote") else "host mailbox")
                            if mode == "decomposed" else ", TTM lattice replicated (one all-gather per step)")
        r1, r3 = price("pass1"), price("pass3")
        dom, other = (r3, r1) if prof["pass3"][1] >= prof["pass1"][1] else (r1, r3)
        n_nbl, ms_nbl = prof["nblist"]
        n_fix, ms_fix = prof["fix_cells"]
        out = {
            "metric": "MD-steps/sec (and atom-steps/sec) for 10M-atom EAM-Al+TTM at 1/2/4/8 GPU",
            "value": args.steps / dt, "unit": "MD-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "atom_steps_per_s": natoms * world * args.steps / dt, "atoms_total": natoms * world,
            "config": {"workload": "FCC Al %dx%dx%d unit cells per GPU (%d atoms/GPU), synthetic tabulated EAM "
                                   "(r_c 6.5 A%s), nbl_margin 0.5, dt 0.1, T0 %.4f eV, %s" %
                                   (args.box[0], args.box[1], args.box[2], natoms,
                                    "; TWO atom types on the L1_2 decoration, 4-column tables with per-column grids" if args.types == 2 else
                                    ("; phi, rho, F on three different grids" if args.tables == "distinct" else ""), args.temp,
                                    "NVE, no TTM" if args.no_ttm else ("3D TTM fd_ext %s, diff_substeps 100, calc_ttm = update_fd, do_ADV %s, activation, FILLMESH, "
                                                                       "100 x (do_DIFF, FILLMESH, ghosts)" % (" ".join(str(v) for v in sysm.fd_ext), "on" if adv_on(args) else "off")) +
                                    (", slab: open x with vacuum on both sides" if args.slab else "") +
                                    (", Beer-Lambert laser source (laser_rescale_ttm)" if args.laser == "beer" else
                                     ", 1D Maxwell solver as laser source (do_fdtd; slab, open x, one row of FD cells%s)" %
                                     (", fitDL from tricubic tables" if args.dl_tables else "") if args.laser == "fdtd" else "")),
                       "atoms_per_gpu": natoms, "cells": list(gdim), "half_list_entries_per_atom": M,
                       "max_neighbours": maxn, "list_rebuilds_in_timed_region": rebuilds,
                       "kernel_family": "lds_tiles" if sysm.get_info("list_mode") == 1 else "global_gather",
                       "tile": [sysm.get_info("tile_x"), sysm.get_info("tile_y"), sysm.get_info("tile_z")],
                       "tile_info": {k: sysm.get_info(k) for k in ("ntiles", "tile_cap", "slot_cap", "max_halo",
                                                                   "tile_threads", "lds_pass1", "lds_pass3", "W")},
                       "parallelism": ("%dx%dx%d ranks, 1 rank/GPU, direct 26-neighbour ghost-cell halo, native step (%s)%s"
                                       % (tuple(cpu_dim) + (md.ops.transport, lattice_note))) if md is not None else "single GPU",
                       # one entry per rank: PCI bus id, UUID, name of its GPU -- N ranks must show N different devices
                       "devices": idents, "distinct_devices": len({i.split(": ", 1)[1].split(" ")[0] for i in idents})},
            # what RCCL reports: the smallest ncclCommCount over the communicators the library created on any rank (0 = no RCCL
            # communicator: single GPU without --force-distributed, or another transport), and that figure per rank
            "rccl_ranks": min(r["rccl_ranks"] for r in rccls), "rccl": rccls,
            "roofline": dom, "roofline_other": other, "roofline_unit": dom.get("unit_busy"), "experiments": sysm.get_info("experiments"),
            "kernel_ms_per_step": {k: v[1] / args.steps for k, v in prof.items() if v[0]},
            "rebuild_ms": {"rebuilds": rebuilds, "make_nblist_ms_each": ms_nbl / max(n_nbl, 1), "fix_cells_ms_each": ms_fix / max(n_fix, 1),
                           "amortised_ms_per_step": (ms_nbl + ms_fix) / args.steps},
            "energy": {"epot_per_atom": sums["tot_pot_energy"] / natoms, "ekin_per_atom": sums["tot_kin_energy"] / natoms},
        }
        if lattice_trial is not None:
            out["lattice_trial"] = dict(lattice_trial, chosen=mode)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
            out["cpu_baseline"]["gpu_over_cpu_atom_steps"] = out["atom_steps_per_s"] / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    _phase(args, rank, "done")
    sysm.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
