#!/bin/bash
# Single-pass attention: per-head rotation of the wave -> sub-tile map and query prefetch (A/B), after the slimmer softmax.
set -u
out=gpurun_out/r2_call20
mkdir -p $out
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $out/smoke.txt; exit 1; }
echo "smoke ok: $(tail -1 $out/smoke.txt)"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > $out/ops_attention.txt 2>&1; rc=$?
echo "ops attention rc=$rc $(tail -1 $out/ops_attention.txt)" | tee -a $out/summary.txt
[ $rc -ne 0 ] && { grep -E "^E|FAILED|Error" $out/ops_attention.txt | head -20; exit 1; }
timeout 200 python scripts/attn_stream_ab.py > $out/attention_variants.jsonl 2> $out/attention_variants.err
cat $out/attention_variants.jsonl | tee -a $out/summary.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], d['parity'], {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
