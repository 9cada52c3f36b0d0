mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_judged_size.py -m gpu -q -x -s > gpurun_out/r05/pytest_js.txt 2>&1; echo "pytest rc $?"
python tools/host_overhead.py > gpurun_out/r05/host_overhead.txt 2>&1; echo "host rc $?"
bash tools/collect_round.sh stats > gpurun_out/r05/collect_stats.log 2>&1; echo "stats rc $?"
