#!/bin/bash
# The headline step under the stream schedules bench.py offers (profiles/r06_schedule_variants.txt): default, sub-batches per modality, one stream.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/sched
Q="--no-cpu-baseline --no-sustained --no-large --no-ladder --no-parity --no-roofline --e2e 0 --steps 20 --warmup 5"
for round in 1 2; do
for v in "" "--split 2 --split-mods v" "--split 2 --split-mods a" "--split 2 --split-mods at" "--streams 0"; do
  timeout 200 python bench.py $Q $v > gpurun_out/sched/x.json 2>> gpurun_out/sched/err.log
  python -c "
import json,sys; x=json.load(open('gpurun_out/sched/x.json')); print(repr(sys.argv[1]).ljust(30), x['value'], x['ms_per_step'], x['config'].get('sub_batches'))" "$v"
done; done | tee gpurun_out/sched/ab.txt
