# A/B of one tuning knob (ppqc_set_tuning through bench.py --tune) on a bench leg, alternating runs in one call (one box).
# usage (GPU box, repo root): bash tools/ab_tune.sh KNOB "v1 v2 ..." [reps] [bench.py arguments, e.g. --fastq --fastq-resident-only]
knob=$1; vals=$2; reps=${3:-3}; shift 3
for rep in $(seq $reps); do for v in $vals; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --tune $knob=$v "$@" > gpurun_out/ab_tune.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_tune.log").read().strip().splitlines()[-1])
print("$knob=$v", d.get("ms_per_step"), d.get("roofline", {}).get("kernel_ms"), d.get("results", ""))
PY
done; done
