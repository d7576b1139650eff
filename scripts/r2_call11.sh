#!/bin/bash
set -u
out=gpurun_out/r2_call11
mkdir -p $out
export TMPDIR=/tmp
timeout 150 python __graft_entry__.py smoke > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc $(tail -1 $out/smoke.log)" | tee $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: smoke failed"; exit 1; }
timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --no-header -p no:cacheprovider -k "attention" > $out/ops.log 2>&1; rc=$?; echo "ops(attention) rc=$rc $(tail -1 $out/ops.log)" | tee -a $out/summary.txt
[[ $rc -ne 0 ]] && { echo "ABORT: attention tests failed"; grep -E "FAILED|Error|assert" $out/ops.log | head -10; exit 1; }
timeout 200 python scripts/attn_stream_ab.py > $out/attn_ab.jsonl 2> $out/attn_ab.err; echo "attn_ab rc=$?" | tee -a $out/summary.txt
cat $out/attn_ab.jsonl | tee -a $out/summary.txt
