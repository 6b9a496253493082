"""GPU test (run with -m gpu): bench.py prints exactly one JSON line with the fields of the driver's contract
(small workload; the reference binary or the oracle port as cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--box", "24", "24", "24", "--steps", "12", "--warmup", "3",
                        "--cpu-box", "24", "--cpu-steps", "3", "--cpu-cells-per-rank", "12", "--cpu-steps-mpi", "3"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3
    assert d["unit"] == "MD-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert abs(d["value"] * d["ms_per_step"] - 1000.0) < 1e-6 * 1000.0
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    if cb["cores"] > 1:          # the reference's MPI + TTM build on the host cores; the one-core NVE figure rides along
        assert "imd_mpi_eam_ttm3d_nbl" in cb["sample"] and cb["serial"]["cores"] == 1 and cb["serial"]["value"] > 0
        assert "do_ADV on" in cb["sample"] and cb["cores_available"] >= cb["cores"] and "cores_cgroup_quota" in cb
    assert "do_ADV on" in d["config"]["workload"]                      # the reference's own calc_ttm sequence (imd_ttm_3D.c:57-60)
    assert "roofline_other" in d and "rebuild_ms" in d


def _one_json_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout          # RCCL's start-up banner and the launcher's chatter must not reach stdout
    return json.loads(lines[0])


def test_bench_spawns_its_ranks_and_refuses_a_mismatch():
    """`python bench.py --gpus N` without a launcher starts the N ranks itself (here 2 ranks sharing the one GPU
    over gloo, the native C step with the host transport) and prints rank 0's line; --gpus that contradicts WORLD_SIZE is
    refused."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--box", "24", "24", "24",
           "--steps", "8", "--warmup", "2", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    d = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600))
    assert d["n_gpus"] == 2 and d["atoms_total"] == 2 * d["config"]["atoms_per_gpu"] and "cpu_baseline" not in d
    assert "2x1x1 ranks" in d["config"]["parallelism"] and "native step" in d["config"]["parallelism"]
    # several ranks: both lattice modes are tried before the warm-up and the faster one that works everywhere runs the benchmark
    lt = d["lattice_trial"]
    assert lt["decomposed"]["ok"] and lt["replicated"]["ok"] and lt["chosen"] in ("decomposed", "replicated")
    assert lt["decomposed"]["faces"].startswith("ghost records in the neighbours' device memory")
    assert ("TTM lattice " + lt["chosen"]) in d["config"]["parallelism"]
    # one rank cannot set its ghost-record buffer up: EVERY rank learns it and the decomposed lattice stays on the host mailbox
    d2 = _one_json_line(subprocess.run(cmd + ["--ttm-lattice", "decomposed", "--hook-remote-fail", "2"], cwd=ROOT, env=env,
                                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600))
    assert "faces: host mailbox" in d2["config"]["parallelism"] and d2["value"] > 0 and "lattice_trial" not in d2
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_bench_nccl_code_path_on_one_rank():
    """the nccl (= RCCL) code path on the one-GPU box -- a one-rank torch.distributed process group
    on the nccl backend (device-tensor all-reduces, barrier, the broadcast of the RCCL ids), the library's own RCCL
    communicators, and with IMDGPU_DIST_SELF_COMM the 26 halo messages / flag all-reduce / lattice all-gather as real RCCL calls"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-distributed", "--init-dist", "--backend", "nccl", "--box", "24", "24", "24",
           "--steps", "8", "--warmup", "2", "--no-cpu-baseline"]
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, IMDGPU_DIST_SELF_COMM="1")
    d = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600))
    assert d["n_gpus"] == 1 and "(rccl)" in d["config"]["parallelism"] and d["value"] > 0


# ---- the bench modes README / DESIGN quote numbers for: each on a small box, one JSON line, the contract
# fields, and the kernels the mode is supposed to run ---------------------------------------------------------------------------
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "roofline_other", "kernel_ms_per_step", "rebuild_ms", "experiments")


def _mode(flags, box=(32, 32, 32), steps=10, warmup=4, env=None, timeout=600):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--box"] + [str(b) for b in box] + ["--steps", str(steps), "--warmup", str(warmup),
                                                                                              "--no-cpu-baseline"] + list(flags)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    d = _one_json_line(subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout))
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] > 0 and d["experiments"] == 0 and d["steps"] == steps and d["n_gpus"] == 1
    assert d["unit"] == "MD-steps/s" and d["dtype"] == "f64" and "workload" in d["config"]
    return d


def test_bench_mode_slab():
    """config 4's geometry on one GPU: open x with vacuum; the persistent sub-step kernel on the filled layers + coefficient force pass"""
    d = _mode(["--slab"])
    assert "slab: open x" in d["config"]["workload"] and d["config"]["cells"][0] > d["config"]["cells"][1]
    assert d["config"]["kernel_family"] == "lds_tiles"
    names = d["roofline"]["kernel"] + d["roofline_other"]["kernel"]
    assert "k_pass1_tile_merged<true, true" in names and "k_pass3_tile_coef" in names
    assert "ttm_diff" in d["kernel_ms_per_step"]


@pytest.mark.parametrize("dl", [False, True])
def test_bench_mode_maxwell_solver(dl):
    """config 5: the 1D Maxwell solver as laser source on a slab (one row of FD cells); --dl-tables: fitDL from tricubic tables"""
    d = _mode(["--laser", "fdtd"] + (["--dl-tables"] if dl else []), steps=8, warmup=4)
    w = d["config"]["workload"]
    assert "1D Maxwell solver" in w and ("fitDL from tricubic tables" in w) == dl and "do_ADV off" in w
    assert d["kernel_ms_per_step"]["ttm_diff"] > 0                    # the solver + diffusion loop ran in the timed region
    assert "k_pass3_tile" in d["roofline"]["kernel"] + d["roofline_other"]["kernel"]


def test_bench_mode_halo_overlap_on_one_rank():
    """pass 1 from a tile queue in two launches under the position halo (option halo_overlap), one rank with all 26 segments
    through RCCL; the line says how many passes took the queue form"""
    d = _mode(["--halo-overlap", "8", "--force-distributed", "--init-dist", "--backend", "nccl"], env=dict(IMDGPU_DIST_SELF_COMM="1"))
    p = d["config"]["parallelism"]
    assert "(rccl)" in p and "position halo under interior tiles of pass 1" in p
    assert "true, false>" in d["roofline"]["kernel"] + d["roofline_other"]["kernel"]            # k_pass1_tile_merged<.., DYN = true, DG = false>
    assert d["kernel_ms_per_step"].get("pass1b", 0) > 0


def test_bench_mode_decomposed_lattice_on_one_rank():
    d = _mode(["--ttm-lattice", "decomposed", "--force-distributed"], env=dict(IMDGPU_DIST_SELF_COMM="1"))
    p = d["config"]["parallelism"]
    assert "native step" in p and "TTM lattice decomposed" in p and "1x1x1 ranks" in p


def test_bench_mode_distinct_grids():
    """one atom type, phi / rho / F on three different grids: the distinct-grids pass 1, the coefficient force pass on rho's grid"""
    d = _mode(["--tables", "distinct"])
    assert "three different grids" in d["config"]["workload"]
    names = d["roofline"]["kernel"] + " " + d["roofline_other"]["kernel"]
    assert "false, true>" in names and "k_pass3_tile_coef" in names                              # k_pass1_tile_merged<.., DYN = false, DG = true>


def test_bench_mode_two_types():
    """two atom types with full-size tables (2 x 4 columns x 4002 rows): the record kernels of the tile family, no global gather"""
    d = _mode(["--types", "2"])
    assert "TWO atom types" in d["config"]["workload"] and d["config"]["kernel_family"] == "lds_tiles"
    names = d["roofline"]["kernel"] + " " + d["roofline_other"]["kernel"]
    assert "k_pass1_tile_rec<true" in names and "k_pass3_tile_rec_lean<" in names             # _lean: the form that leaves the lattice chain its registers
    assert "pass2" not in d["kernel_ms_per_step"]                     # the embedding pass is folded into pass 1


def test_bench_reports_rccl_ranks_and_devices():
    """what RCCL itself says (ncclCommCount of every communicator the library created) and every rank's GPU
    (PCI bus id, UUID) are in the line -- here one rank with every message through RCCL: rccl_ranks 1 on one device; without the
    multi-rank layer no communicator exists and rccl_ranks is 0"""
    d = _mode(["--force-distributed", "--init-dist", "--backend", "nccl"], env=dict(IMDGPU_DIST_SELF_COMM="1"))
    assert d["rccl_ranks"] == 1 and d["rccl"][0]["rccl_comms"] >= 2 and d["rccl"][0]["rccl_device"] == 0
    assert len(d["config"]["devices"]) == 1 and "uuid" in d["config"]["devices"][0] and d["config"]["distinct_devices"] == 1
    d0 = _mode([])
    assert d0["rccl_ranks"] == 0 and d0["config"]["devices"][0].startswith("rank 0: ")


def test_transports_run_between_the_persistent_launches():
    """the default multi-rank step issues every transport (dF halo, lattice exchange, position halo, flag
    all-reduce) on the main stream BETWEEN the persistent force launches, or on the lattice stream in front of an event the
    force pass waits for -- no RCCL kernel ever has to be placed beside a pass that holds every CU.  Asserted on the device
    clock (IMDGPU_TIMELINE marks of a one-rank run with every message through RCCL): begin <= pass1 <= pass2 <= dF_halo <=
    pass3 <= move <= halo+flag+pass1ahead in every step, and the mark set is exactly the documented one."""
    import re
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-distributed", "--init-dist", "--backend", "nccl", "--ttm-lattice", "decomposed",
           "--box", "24", "24", "24", "--steps", "6", "--warmup", "3", "--no-cpu-baseline"]
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, IMDGPU_DIST_SELF_COMM="1", IMDGPU_TIMELINE="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    _one_json_line(r)
    steps = [l for l in r.stderr.splitlines() if l.startswith("[imdgpu timeline] step")]
    assert len(steps) >= 6
    seen = 0
    for l in steps:
        seq = [(k, float(v)) for k, v in re.findall(r"(\S+)=([0-9.]+)", l)]
        names = [k for k, _ in seq]
        if "dF_halo" not in names:
            continue                                   # a step that began with a rebuild prints the rebuild's marks
        seen += 1
        # marks of the MAIN stream in issue order (every call of the force launcher marks "pass1" / "pass3" around what it launched;
        # the ttm_* marks belong to the lattice stream): their device times never go backwards ...
        main = [(k, t) for k, t in seq if not k.startswith("ttm_")]
        assert [t for _, t in main] == sorted(t for _, t in main), l
        # ... and the order of issue is: embedding pass / dF halo (an RCCL group) / EAM force pass / integrator / position halo +
        # flag all-reduce + the guarded pass 1 of the next step
        mn = [k for k, _ in main]
        i_df = mn.index("dF_halo"); i_mv = mn.index("move"); i_end = mn.index("halo+flag+pass1ahead")
        assert mn.index("pass2") < i_df < i_mv < i_end, l
        assert mn[i_df + 1: i_mv] == ["pass1", "pass3"], l           # the force launcher's phase 3: nothing else between the dF halo and the integrator
        assert "pass1" in mn[i_mv + 1: i_end], l                     # the next step's pass 1 behind halo and flags
    assert seen >= 3


def test_bench_watchdog_reports_a_rank_that_hangs():
    """a rank that never arrives (test hook: rank 1 sleeps in front of the warm-up) -- every rank's watchdog
    ends its process with a status, nothing is re-executed, the parent exits non-zero and prints each rank's last phase"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--box", "16", "16", "16",
           "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--ttm-lattice", "replicated", "--hook-rank-hang", "1", "--rank-timeout", "45"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert "last status of every rank" in r.stderr and "rank 1: WATCHDOG" in r.stderr and "warm-up" in r.stderr
