#!/bin/bash
set -u
out=gpurun_out/r2_call13
mkdir -p $out
timeout 200 python scripts/bench_segbias.py > $out/segbias.jsonl 2> $out/segbias.err; echo "segbias rc=$?" | tee $out/summary.txt
cat $out/segbias.jsonl | tee -a $out/summary.txt
