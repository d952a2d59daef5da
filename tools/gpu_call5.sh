#!/bin/bash
# round 2, call 5: stage-2 text / fusion tower parity (row kernels, tower vs the reference's fixture, BERT-large vs the oracle)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bert_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/call5_bert.log
cat gpurun_out/call5_bert.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k bert_large 2>&1 | tail -30 > gpurun_out/call5_bert_large.log
cat gpurun_out/call5_bert_large.log
