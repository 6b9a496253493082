set -o pipefail
mkdir -p gpurun_out/r5f
timeout -k 10 1100 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_ttm_persist.py -x -q -m gpu > gpurun_out/r5f/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5f/tests.log
tail -25 gpurun_out/r5f/tests.log
