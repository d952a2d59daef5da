cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gpu.py -m gpu -q -s -k "delayed or 6B_encoder or requantises or nan" 2>&1 | tail -30 | cut -c1-1500
timeout 600 python bench.py --model 6B --batch 16 --fp8 --fp8-scaling delayed --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_6b_fp8_delayed.json 2> gpurun_out/r3_6b_fp8_delayed.err; cut -c1-300 gpurun_out/r3_6b_fp8_delayed.json; tail -2 gpurun_out/r3_6b_fp8_delayed.err
