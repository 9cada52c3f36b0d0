mkdir -p gpurun_out/r05
python tools/cfg3_vs_oracle.py --reuse-ld tests/golden/cfg3_extended_precision_iterates.npz --refine 2 --xcorr 0,e512,e256,2 --no-f64-oracle 200 500 1000 2000 > gpurun_out/r05/cfg3_xcorr_f.txt 2> gpurun_out/r05/cfg3_xcorr_f.err; echo "tool rc $?"
B="python bench.py --no-cpu-baseline --no-cfg4-strong --no-baseline-configs --no-single-process --no-general-path --no-alt-storage --no-concurrent --steps 20 --warmup 3"
for x in 0 e512 e256; do LPVS_XUPDATE_CORRECTION=$x $B > gpurun_out/r05/bench_xcorrf_$x.json 2> gpurun_out/r05/bench_xcorrf_$x.err; echo "bench $x rc $?"; done
python -m pytest tests -m gpu -x -q --durations=60 > gpurun_out/r05/pytest_full.txt 2>&1; echo "pytest rc $?"
