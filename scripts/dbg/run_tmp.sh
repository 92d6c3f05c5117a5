set -x
mkdir -p gpurun_out/s5
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/s5/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s5/pytest_gpu.log
tail -4 gpurun_out/s5/pytest_gpu.log
