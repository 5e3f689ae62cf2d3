# A/B of one environment knob on the streamed (pinned host -> GPU) pipeline, alternating runs in one call.
# usage (GPU box, repo root): bash tools/ab_stream.sh VAR "v1 v2 ..." [reps]
var=$1; vals=$2; reps=${3:-3}
for rep in $(seq $reps); do for v in $vals; do
  export $var=$v
  timeout -k 10 300 python bench.py --stream --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_stream_run.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_stream_run.log").read().strip().splitlines()[-1])
print("$var=$v", d["metric"], d["value"], "ms_per_step", d["ms_per_step"])
PY
done; done
