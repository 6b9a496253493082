set -o pipefail
mkdir -p gpurun_out/r5m
timeout -k 10 1150 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5m/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5m/tests.log
tail -6 gpurun_out/r5m/tests.log
