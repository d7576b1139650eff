#!/bin/bash
# rocprofv3 kernel stats of the audio-only step at batch 32 (BASELINE configs[1]) and of the large trio
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c23; mkdir -p "$O"
export TMPDIR=/tmp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof_a" -o step -- python "$OLDPWD/bench.py" --modalities a --batch 32 --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --no-large --no-sustained --e2e 0 --streams 0 > /dev/null 2>&1; echo "prof audio rc=$?")
f=$(find "$O/prof_a" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" "$O/audio_b32_kernel_stats.csv"; rm -rf "$O/prof_a"; head -12 "$O/audio_b32_kernel_stats.csv" | cut -c1-150
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof_l" -o step -- python "$OLDPWD/bench.py" --config large --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-roofline --no-sustained --e2e 0 --streams 0 > /dev/null 2>&1; echo "prof large rc=$?")
f=$(find "$O/prof_l" -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" "$O/large_kernel_stats.csv"; rm -rf "$O/prof_l"; head -8 "$O/large_kernel_stats.csv" | cut -c1-150
