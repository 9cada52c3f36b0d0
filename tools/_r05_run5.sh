mkdir -p gpurun_out/r05
OMP_NUM_THREADS=16 python tools/cfg3_vs_oracle.py --reuse-ld tests/golden/cfg3_extended_precision_iterates.npz --save gpurun_out/r05/cfg3_fixture.npz 200 500 1000 2000 > gpurun_out/r05/cfg3_default.txt 2> gpurun_out/r05/cfg3_default.err; echo "tool rc $?"
python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "structured_vs_dense" > gpurun_out/r05/pytest_pair.txt 2>&1; echo "pair rc $?"
python -m pytest tests -m gpu -q --durations=40 > gpurun_out/r05/pytest_full2.txt 2>&1; echo "pytest rc $?"
