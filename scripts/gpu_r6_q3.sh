#!/bin/bash
# round 6, call 3: lean 16-bit epilogue (epi0_block) in gemm16p and gemm16q: bits (pytest), timing A/B, timelines
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6q3; mkdir -p $O/p $O/q1 $O/q2
timeout 900 python -m pytest tests/test_gemm16q_gpu.py tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm16" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 scripts/probes/gemm16_bench.bin 20 20 clip > $O/bench_clip.jsonl 2>&1; echo "clip rc=$?"
timeout 300 scripts/probes/gemm16_bench.bin 20 20 hubert > $O/bench_hubert.jsonl 2>&1; echo "hubert rc=$?"
timeout 300 scripts/probes/gemm16_bench.bin 20 20 roberta > $O/bench_roberta.jsonl 2>&1; echo "roberta rc=$?"
grep differing $O/bench_*.jsonl | grep -v '"differing_words": 0}' | grep -v "hubert fc1\|conv1-like" | head
MER_NO_Q=1 MER_STAMP=$O/p timeout 200 scripts/probes/gemm16_bench.bin 5 5 clip > $O/p.jsonl 2>&1; echo "p rc=$?"
MER_NO_Q=1 MER_STAMP=$O/q1 MER_STAMP_Q=1 timeout 200 scripts/probes/gemm16_bench.bin 5 5 clip > $O/q1.jsonl 2>&1; echo "q1 rc=$?"
MER_NO_Q=1 MER_STAMP=$O/q2 MER_STAMP_Q=2 timeout 200 scripts/probes/gemm16_bench.bin 5 5 clip > $O/q2.jsonl 2>&1; echo "q2 rc=$?"
python scripts/gemm16p_timeline.py $O/p/stamps_0*.bin > $O/timeline_p.txt 2>&1
python scripts/gemm16p_timeline.py --q=2 $O/q1/stamps_0*.bin > $O/timeline_q1.txt 2>&1
python scripts/gemm16p_timeline.py --q=4 $O/q2/stamps_0*.bin > $O/timeline_q2.txt 2>&1
rm -rf $O/p $O/q1 $O/q2
