#!/bin/bash
# r2 call 34: grouped end-of-backward weight gradients of the text tower -- test + stage-2 step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_bert_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/call34_tests.log; cat $O/call34_tests.log
timeout 400 python tools/bench_stage2.py --batch-text --graph --steps 10 --warmup 3 > $O/call34_stage2_graph.json 2> $O/call34_a.err; cut -c1-260 $O/call34_stage2_graph.json; tail -2 $O/call34_a.err
timeout 400 python tools/bench_stage2.py --batch-text --graph --group-wgrad --steps 10 --warmup 3 > $O/call34_stage2_graph_gw.json 2> $O/call34_b.err; cut -c1-260 $O/call34_stage2_graph_gw.json; tail -2 $O/call34_b.err
timeout 400 python tools/bench_stage2.py --batch-text --group-wgrad --steps 6 --warmup 2 > $O/call34_stage2_eager_gw.json 2> $O/call34_c.err; cut -c1-260 $O/call34_stage2_eager_gw.json; tail -2 $O/call34_c.err
