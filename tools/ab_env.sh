# A/B of one environment knob on the whole bench step, alternating runs in one call (one box).
# usage (GPU box, repo root): bash tools/ab_env.sh VAR "v1 v2 ..." [reps]
var=$1; vals=$2; reps=${3:-3}
for rep in $(seq $reps); do for v in $vals; do
  export $var=$v
  timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-h2d > gpurun_out/ab_env.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_env.log").read().strip().splitlines()[-1])
print("$var=$v ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "n50", d["config"]["n50"])
PY
done; done
