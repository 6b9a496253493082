set -o pipefail
mkdir -p gpurun_out/r5d
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5d/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5d/tests.log
tail -15 gpurun_out/r5d/tests.log
