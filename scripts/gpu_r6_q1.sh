#!/bin/bash
# round 6, call 1: gemm16q (lagged groups) bits + timing against gemm16p on the bench's GEMM shapes, then its pytest file
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6q1; mkdir -p $O
timeout 240 scripts/probes/gemm16_bench.bin 10 10 clip > $O/bench_clip.jsonl 2>&1; echo "clip rc=$?"
grep -c '"differing_words": 0}' $O/bench_clip.jsonl; grep differing $O/bench_clip.jsonl | grep -v '"differing_words": 0}' | head
grep -E "gemm16p|gemm16q|tile kernel" $O/bench_clip.jsonl | cut -c1-200
timeout 240 scripts/probes/gemm16_bench.bin 10 10 hubert > $O/bench_hubert.jsonl 2>&1; echo "hubert rc=$?"
grep differing $O/bench_hubert.jsonl | grep -v '"differing_words": 0}' | head
grep -E "gemm16p|gemm16q" $O/bench_hubert.jsonl | cut -c1-200
timeout 900 python -m pytest tests/test_gemm16q_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
