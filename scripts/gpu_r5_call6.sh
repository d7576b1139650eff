#!/bin/bash
# round 5, call 6: the test edits made after the evidence suite (norm-free asserts, the outlier conditioning bound, vectorised constant-row pre-check)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c6; mkdir -p "$O"
timeout 900 python -m pytest tests/test_encoders_gpu.py tests/test_parity_hardening_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "hubert_base_5s or clip_base16_8frames or roberta_base_64tok or activation_outliers_post_ln or noise_clip_among" > "$O/t.log" 2>&1; echo "tests rc=$?"
grep -E "\[mean\]|per clip under accurate|silence|passed|failed|Error|assert" "$O/t.log" | grep -v Warning | cut -c1-400 | tail -30
