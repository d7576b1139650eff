#!/bin/bash
# round 5, call 8: the live-HF file with seeded modules and the self-check's 20 % margin; every test that depends on a self-check decision
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c8; mkdir -p "$O"
timeout 900 python -m pytest tests/test_from_hf_gpu.py tests/test_encoders_gpu.py tests/test_affectgpt.py -m gpu -q --no-header -p no:cacheprovider -s -k "from_hf or by_name or activation_outliers or affectgpt" > "$O/t.log" 2>&1; echo "tests rc=$?"
grep -E "self-check|from_hf\[|vs the live|by name \[|default constructor|passed|failed|Error|assert" "$O/t.log" | grep -v Warning | cut -c1-260 | tail -50
