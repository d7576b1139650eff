#!/bin/bash
# round 6, call 2: timelines of gemm16q (D2 S1, D2 S2) against gemm16p on CLIP's four block GEMMs
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6q2; mkdir -p $O/p $O/q1 $O/q2
MER_NO_Q=1 MER_STAMP=$O/p timeout 200 scripts/probes/gemm16_bench.bin 5 5 clip > $O/p.jsonl 2>&1; echo "p rc=$?"
MER_NO_Q=1 MER_STAMP=$O/q1 MER_STAMP_Q=1 timeout 200 scripts/probes/gemm16_bench.bin 5 5 clip > $O/q1.jsonl 2>&1; echo "q1 rc=$?"
MER_NO_Q=1 MER_STAMP=$O/q2 MER_STAMP_Q=2 timeout 200 scripts/probes/gemm16_bench.bin 5 5 clip > $O/q2.jsonl 2>&1; echo "q2 rc=$?"
python scripts/gemm16p_timeline.py $O/p/stamps_0*.bin > $O/timeline_p.txt 2>&1
python scripts/gemm16p_timeline.py --q=2 $O/q1/stamps_0*.bin > $O/timeline_q1.txt 2>&1
python scripts/gemm16p_timeline.py --q=4 $O/q2/stamps_0*.bin > $O/timeline_q2.txt 2>&1
rm -rf $O/p $O/q1 $O/q2
tail -5 $O/timeline_q1.txt
