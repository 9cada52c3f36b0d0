mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r05/smoke.txt 2>&1; echo "smoke rc $?"
python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/r05/pytest_final.txt 2>&1; echo "pytest rc $?"
S=$(date +%s); python bench.py > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err; echo "bench rc $? wall $(( $(date +%s) - S )) s"
