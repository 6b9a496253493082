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
