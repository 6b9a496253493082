#!/usr/bin/env python3
"""Throughput of the LEVEL-2 drop-in (IMD_SHIM_LEVEL=2: the reference's main loop with the atoms resident on the device,
host cells refreshed only where the host reads them) beside level 1 and beside bench.py's device-resident loop on the same
box: python tools/level2_timing.py [unit cells] [ttm 0|1]

WALL time per MD step = (wall of a long run - wall of a short run) / (difference of steps): set-up cancels.  No energy file
and no checkpoint between the first and the last step (eng_int / checkpt_int beyond the run): "between I/O steps"."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imd_amd import potential, ttm as ttm_mod  # noqa: E402
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ttm = len(sys.argv) > 2 and sys.argv[2] == "1"
PARAM = """coordname _fcc
outfiles out
ensemble %s
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
TTM = """fd_ext 2 2 2
fd_n_timesteps 100
fd_min_atoms 5
atomic_weight 26.9815
atomic_charge 13
ttm_int 100000
I0 1.0
lambda 800e-9
laser_sigma_t 100e-15
laser_t_0 1.0
"""
exe = "imd_mpi_eam_ttm3d_nbl_gpu" if ttm else "imd_eam_nve_nbl_gpu"
short, long_ = (20, 120)
res = {}
for level in ("1", "2"):
    wall = {}
    for steps in (short, long_) if level == "2" else (10, 30):
        with tempfile.TemporaryDirectory() as d:
            potential.write_synthetic_eam_files(d)
            if ttm:
                cve, ee = ttm_mod.synthetic_eos()
                ttm_mod.write_eos_table(os.path.join(d, "EOS_cve_from_r_te.txt"), cve)
                ttm_mod.write_eos_table(os.path.join(d, "EOS_ee_from_r_tesqrt.txt"), ee)
            open(os.path.join(d, "t.param"), "w").write(PARAM % ("ttm" if ttm else "nve", steps, nb, nb, nb) + (TTM if ttm else ""))
            env = dict(os.environ, IMD_PROBE_STEPS="-1", IMD_SHIM_NO_NBL_EXPORT="1", OMP_WAIT_POLICY="passive", IMD_SHIM_LEVEL=level)
            t0 = time.time()
            r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), "-p", "t.param"], cwd=d, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            wall[steps] = time.time() - t0
            assert r.returncode == 0, r.stdout[-2000:]
    a, b = sorted(wall)
    res[level] = (wall[b] - wall[a]) / (b - a)
    print("level %s drop-in (%s): %8.3f ms wall per MD step (%d atoms)" % (level, exe, 1e3 * res[level], 4 * nb ** 3), flush=True)
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--box", str(nb), str(nb), str(nb), "--steps", "100", "--warmup", "20",
                    "--no-cpu-baseline", "--no-adv"] + ([] if ttm else ["--no-ttm"]), stdout=subprocess.PIPE, text=True)
b = json.loads(r.stdout.strip().splitlines()[-1])
print("bench.py --box %d %d %d%s: %8.3f ms per MD step" % (nb, nb, nb, "" if ttm else " --no-ttm", b["ms_per_step"]))
print(json.dumps(dict(atoms=4 * nb ** 3, ttm=ttm, level1_ms=1e3 * res["1"], level2_ms=1e3 * res["2"], bench_ms=b["ms_per_step"],
                      level2_over_bench=1e3 * res["2"] / b["ms_per_step"], level1_over_level2=res["1"] / res["2"])))
