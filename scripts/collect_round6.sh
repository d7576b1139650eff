#!/bin/bash
# Copies what `gpurun ... bash scripts/gpu_round6.sh <tag>` merged into gpurun_out/<tag>/ to profiles/r06_* and stamps the rocprofv3 summaries
# with the sha of the library sources in the tree (scripts/stamp_kernel_stats.py); run it HERE, on the tree the call was made from.
set -e
tag=${1:-r06}; O=gpurun_out/$tag
cp $O/pmc_hbm_traffic.json profiles/r06_pmc_hbm_traffic.json; cp $O/pmc_hbm_traffic.txt profiles/r06_pmc_hbm_traffic.txt
cp $O/pmc_mfma.json profiles/r06_pmc_mfma.json; cp $O/pmc_mfma.txt profiles/r06_pmc_mfma.txt
cp $O/bench.json profiles/r06_bench_default.json; cp $O/bench_with_pmc.json profiles/r06_bench_default_with_pmc.json
for n in audio_b32 audio_b64 visual_b64 text_b64 mx balanced; do cp $O/bench_$n.json profiles/r06_bench_$n.json; done
cp $O/parity_lines.txt profiles/r06_parity_lines.txt; cp $O/abi_selftest.jsonl profiles/r06_abi_selftest.jsonl
cp $O/load_time_ladder.json profiles/r06_load_time_ladder.json
grep -E '"variant"|"check"' $O/gemm16_bench.jsonl > profiles/r06_gemm16_bench_random_operands.jsonl
python scripts/stamp_kernel_stats.py $O/headline_kernel_stats.csv profiles/r06_bench_mean_b64
python scripts/stamp_kernel_stats.py $O/audio_b32_kernel_stats.csv profiles/r06_bench_audio_b32 "rocprofv3 --kernel-trace --stats -- python bench.py --modalities a --batch 32 --steps 8 --warmup 2 --streams 0"
python scripts/stamp_kernel_stats.py $O/large_kernel_stats.csv profiles/r06_bench_large "rocprofv3 --kernel-trace --stats -- python bench.py --config large --steps 2 --warmup 1 --streams 0"
sha=$(python -c "import json; print(json.load(open('profiles/r06_pmc_mfma.json'))['_source_sha'])")
{ grep -E "passed|failed" $O/suite.log | tail -1; echo "(gpurun call: bash scripts/gpu_round6.sh $tag; kernel_source_sha $sha over mertools_amd/csrc/*)"; } > profiles/r06_gpu_suite_summary.txt
cat profiles/r06_gpu_suite_summary.txt
