# A/B of one environment knob on the default bench's short end-to-end (pinned host -> GPU) leg, alternating runs.
# usage (GPU box, repo root): bash tools/ab_h2d.sh VAR "v1 v2 ..." [reps]
var=$1; vals=$2; reps=${3:-3}
for rep in $(seq $reps); do for v in $vals; do
  export $var=$v
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_h2d_run.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_h2d_run.log").read().strip().splitlines()[-1])
e=d["config"]["end_to_end_incl_h2d"]
print("$var=$v h2d leg", e["gbases_per_s"], "Gbases/s", e["h2d_GBps"], "GB/s")
PY
done; done
