#!/bin/bash
# r2 call 30: what operand traffic / C stores / CU count cost the power-limited GEMM; row kernels behind the GEMM; wgrad stream A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out; mkdir -p $O
timeout 300 python tools/gemm_traffic_probe.py 3 > $O/call30_traffic_probe.jsonl 2> $O/call30_traffic_probe.err; cut -c1-330 $O/call30_traffic_probe.jsonl; tail -3 $O/call30_traffic_probe.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b32 --no-kernel-events > $O/call30_bench_base.json 2> $O/call30_bench_base.err; cut -c1-200 $O/call30_bench_base.json; tail -2 $O/call30_bench_base.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b32 --no-kernel-events --wgrad-stream > $O/call30_bench_wgs.json 2> $O/call30_bench_wgs.err; cut -c1-200 $O/call30_bench_wgs.json; tail -2 $O/call30_bench_wgs.err
