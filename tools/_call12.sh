cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
IVH_ATTN_PPW=2 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fused_ops_gpu.py -m gpu -q -k "attn or attention" 2>&1 | tail -4
for ppw in 1 2; do echo "== ppw $ppw"; IVH_ATTN_PPW=$ppw timeout 300 python tools/bench_attn.py --quick 2>&1 | grep '"32x32x16"' | grep fwd | cut -c1-200; done
IVH_ATTN_PPW=2 timeout 200 python tools/attn_timeline.py 128 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-700
