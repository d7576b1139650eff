#!/bin/bash
# Multi-row LayerNorm (a wave walks several rows, gamma / beta in LDS, next row prefetched): tests, then the headline bench with and without it.
set -u
out=gpurun_out/r2_call24
mkdir -p $out
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || { echo "smoke failed"; tail -5 $out/smoke.txt; exit 1; }
echo "smoke ok: $(tail -1 $out/smoke.txt)"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layernorm or bert_embed or vit_assemble" > $out/ops_ln.txt 2>&1; rc=$?
echo "ops layernorm rc=$rc $(tail -1 $out/ops_ln.txt)" | tee -a $out/summary.txt
[ $rc -ne 0 ] && { grep -E "^E|FAILED|Error" $out/ops_ln.txt | head -20; exit 1; }
for rows in 1 0; do
  MER_OPTIONS="ln_rows=$rows" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_rows$rows.json 2> $out/bench_rows$rows.err
  echo "ln_rows=$rows rc=$? $(python -c "import json;d=json.load(open('$out/bench_rows$rows.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], d['parity'], {k:(v['ms_share'],v['tflops'] or v['gbps']) for k,v in r['other_kernels'].items() if k in ('gemm16_mx','attention','layernorm')})" 2>/dev/null)" | tee -a $out/summary.txt
done
