#!/bin/bash
# round 5, call 12: what the self-check's rungs cost — the audio-only step and the headline step under mean_conv3 / mean_a2 next to the default
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c12; mkdir -p "$O"
for prec in mean mean_conv3 mean_a2 accurate; do
  timeout 200 python bench.py --modalities a --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/audio_$prec.json" 2>> "$O/bench.err"; echo "audio $prec rc=$?"
done
timeout 200 python bench.py --precision mean_conv3 --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --no-large --e2e 0 > "$O/trimodal_mean_conv3.json" 2>> "$O/bench.err"; echo "trimodal rc=$?"
python - <<'P'
import json,glob
out={}
for f in sorted(glob.glob("gpurun_out/r5c12/*.json")):
    if f.endswith("summary.json"): continue
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    k=f.split("/")[-1][:-5]; out[k]={"clips_per_s":d["value"],"ms_per_step":d["ms_per_step"],"parity":d["parity"],"whole_step_frac":d["roofline"]["whole_step_frac"]}
    print(k, out[k])
json.dump(out, open("gpurun_out/r5c12/summary.json","w"), indent=1)
P
