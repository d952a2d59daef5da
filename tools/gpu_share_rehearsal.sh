#!/bin/bash
# Rehearsal of the N > 1 step on a 1-GPU box: two real processes share cuda:0, collectives over gloo (bench.py --share-gpu).
# Not a throughput measurement -- it checks that the multi-process step (segmented graphs + bucket collectives + all-ranks checks +
# timing protocol) runs to the JSON line, and that the graph-segments and eager modes agree on the loss.
tag=${1:-r3}
out=gpurun_out
mkdir -p $out
rm -f $out/${tag}_share_status.txt
export TMPDIR=/tmp
common="--gpus 2 --share-gpu --steps 4 --warmup 2 --reduce-dtype fp32 --no-cpu-baseline --no-b32 --no-kernel-events --dist-timeout 180"
for mode in auto eager; do
  timeout 900 python bench.py $common --model 1B --batch 16 --dist-mode $mode > $out/${tag}_share_1B_b16_$mode.json 2> $out/${tag}_share_1B_b16_$mode.err
  rc=$?
  echo "1B $mode rc $rc" >> $out/${tag}_share_status.txt
  if [ $rc -ne 0 ]; then grep -v "^\s*$" $out/${tag}_share_1B_b16_$mode.err | grep -v "amdgpu.ids\|socket.cpp" | head -30; cat $out/${tag}_share_status.txt; exit 1; fi
done
timeout 600 python bench.py ${common/--reduce-dtype fp32/--reduce-dtype bf16} --model 1B --batch 16 --dist-mode auto > $out/${tag}_share_1B_b16_auto_bf16wire.json 2> $out/${tag}_share_1B_b16_auto_bf16wire.err
echo "1B auto bf16 wire rc $?" >> $out/${tag}_share_status.txt
cat $out/${tag}_share_status.txt
for f in $out/${tag}_share_1B_b16_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k:d.get(k) for k in ("n_gpus","dist_mode","dist_note","backend","shared_gpu","graph_segments","reduce_buckets","loss","ms_per_step")})
except Exception as e:
    print(sys.argv[1], "no line", e)
P
done
tail -n 5 $out/${tag}_share_1B_b16_*.err
