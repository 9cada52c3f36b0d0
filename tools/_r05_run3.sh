mkdir -p gpurun_out/r05
python tools/cfg3_vs_oracle.py --reuse-ld tests/golden/cfg3_extended_precision_iterates.npz --refine 2 --xcorr 2,4,8,e128,e256,e512 --no-f64-oracle 200 500 1000 2000 > gpurun_out/r05/cfg3_xcorr_sched.txt 2> gpurun_out/r05/cfg3_xcorr_sched.err; echo "tool rc $?"
B="python bench.py --no-cpu-baseline --no-cfg4-strong --no-baseline-configs --no-single-process --no-general-path --no-alt-storage --no-concurrent --steps 20 --warmup 3"
for x in 0 2 4 8 e256 e512; do LPVS_XUPDATE_CORRECTION=$x $B > gpurun_out/r05/bench_xcorr_$x.json 2> gpurun_out/r05/bench_xcorr_$x.err; echo "bench $x rc $?"; done
python tools/cfg5_ab.py > gpurun_out/r05/cfg5_ab2.txt 2>&1; echo "cfg5 rc $?"
python -m pytest tests/test_gpu_regressions.py tests/test_gpu_one_launch.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r05/pytest_c.txt 2>&1; echo "pytest rc $?"
