cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3c10_pytest.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r3c10_pytest.log | cut -c1-220
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c10_bench.json 2> gpurun_out/r3c10_bench.err; echo "bench rc $?"; cut -c1-260 gpurun_out/r3c10_bench.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --residual fp32 --no-b32 --no-kernel-events > gpurun_out/r3c10_bench_fp32.json 2> gpurun_out/r3c10_bench_fp32.err; cut -c1-260 gpurun_out/r3c10_bench_fp32.json
