#!/bin/bash
# round 5, call 1: the live-HF from_hf tests + the f1/f2 tests that gained the default preset
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c1; mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_from_hf_gpu.py -m gpu -q --no-header -p no:cacheprovider -s > "$O/from_hf.log" 2>&1; echo "from_hf rc=$?"
grep -E "from_hf|driver by name|self-check|passed|failed|Error|error" "$O/from_hf.log" | grep -v Warning | tail -60
timeout 900 python -m pytest tests/test_encoders_gpu.py tests/test_dinov2.py -m gpu -q --no-header -p no:cacheprovider -s -k "data2vec or wavlm or electra or clip_large14 or videomae_base or dinov2" > "$O/f2.log" 2>&1; echo "f2 rc=$?"
grep -E "^\S+\[|passed|failed|Error|assert" "$O/f2.log" | grep -v Warning | tail -60
