#!/bin/bash
# round 5, call 2: passes=6 op test, mean_a2 preset, constant-row escalation, outlier ladder, d2v-audio diagnosis
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c2; mkdir -p "$O"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "passes6 or three_pass or persistent_equals or bias_table" > "$O/ops.log" 2>&1; echo "ops rc=$?"; tail -5 "$O/ops.log"
timeout 900 python -m pytest tests/test_encoders_gpu.py tests/test_parity_hardening_gpu.py tests/test_from_hf_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "hubert_base_5s or clip_base16_8frames or roberta_base_64tok or activation_outliers_post_ln or noise_clip_among or audio_driver_by_name" > "$O/enc.log" 2>&1; echo "enc rc=$?"
grep -E "^\S+.*\[|passed|failed|Error|assert|by name" "$O/enc.log" | grep -v Warning | tail -70
timeout 300 python scripts/diag_d2v_audio.py > "$O/d2v.log" 2>&1; echo "diag rc=$?"; grep "d2v-audio\|Error" "$O/d2v.log"
