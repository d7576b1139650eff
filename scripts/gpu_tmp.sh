cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r6d1; mkdir -p $O
Q="--no-cpu-baseline --no-sustained --no-large --no-ladder --no-roofline --e2e 0 --steps 20 --warmup 5"
for round in 1 2 3; do
  for alt in 0 1; do
    for m in avt v a t; do
      MER_OPTIONS=alt_dir=$alt timeout 200 python bench.py $Q --modalities $m > $O/alt${alt}_${m}_$round.json 2>> $O/err.log
      python - "$O/alt${alt}_${m}_$round.json" $alt $m $round <<'P'
import json, sys
x = json.load(open(sys.argv[1])); print("alt_dir", sys.argv[2], sys.argv[3], "round", sys.argv[4], x["value"], x["ms_per_step"], x.get("parity"))
P
    done
  done
done
tail -3 $O/err.log
