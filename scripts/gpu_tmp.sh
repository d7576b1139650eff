cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6t3; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $O/suite.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/suite.log | tail -12
