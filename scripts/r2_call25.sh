#!/bin/bash
# After the multi-row LayerNorm: the encoder tests whose conv front end runs LayerNorm over tens of thousands of rows (large-style HuBERT,
# data2vec-audio, WavLM-large style, the large trio), which is where the new kernel is selected inside a model.
set -u
out=gpurun_out/r2_call25
mkdir -p $out
timeout 130 python -m pytest tests/test_encoders_gpu.py -x -q -m gpu -s -k "ragged or large_trio or data2vec_audio_base or tiny_large_style" > $out/enc.txt 2>&1; rc=$?
echo "enc rc=$rc $(grep -E 'passed|failed' $out/enc.txt | tail -1)" | tee $out/summary.txt
grep -E "^\.?(hubert|large|data2vec)" $out/enc.txt | cut -c1-200 | tee -a $out/summary.txt
if [ $rc -ne 0 ]; then grep -E "^E|FAILED" $out/enc.txt | head; fi
