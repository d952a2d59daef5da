#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for M in 834 53376; do
IVH_QK_W4=0 timeout 200 python tools/diag_qk_bwd.py /tmp/qk0_$M.pt $M > $O/r3c5_qk0_$M.log 2>&1
IVH_QK_W4=1 timeout 200 python tools/diag_qk_bwd.py /tmp/qk1_$M.pt $M > $O/r3c5_qk1_$M.log 2>&1
python - <<PY >> $O/r3c5_qkcmp.log 2>&1
import torch
a, b = torch.load("/tmp/qk0_$M.pt"), torch.load("/tmp/qk1_$M.pt")
for k in a:
    x, y = a[k].float(), b[k].float()
    bad = (~torch.isfinite(y)).nonzero()
    print($M, k, "rel", ((x - y).norm() / x.norm()).item(), "nonfinite", bad.shape[0], bad[:4].tolist())
    if k == "dd":
        diff = (x - y).abs().amax(1)
        rows = (diff > 0.05).nonzero().flatten()
        print("  rows differing", rows.numel(), rows[:12].tolist(), rows[-4:].tolist())
PY
done
cat $O/r3c5_qk*_*.log $O/r3c5_qkcmp.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r3c5_bench.json 2> $O/r3c5_bench.err; echo "bench rc $?"
cut -c1-260 $O/r3c5_bench.json
timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k "1B_graph_replayed or stage2_1B_vision" 2>&1 | tail -4
