cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 200 python tools/attn_timeline.py 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_attn_fwd_timeline_b128.md
timeout 200 python tools/attn_timeline.py 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_attn_fwd_timeline_b32.md
timeout 300 python tools/bench_attn.py --quick 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_attn_bench.jsonl
IVH_ATTN_DEFER=1 timeout 300 python tools/bench_attn.py --quick 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_attn_bench_defer.jsonl
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "flash_attn" 2>&1 | tail -3
