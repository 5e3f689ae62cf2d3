# A/B of one environment knob on the resident FASTQ leg, alternating runs in one call (one box).
# usage (GPU box, repo root): bash tools/ab_fastq_env.sh VAR "v1 v2 ..." [reps]
var=$1; vals=$2; reps=${3:-3}
for rep in $(seq $reps); do for v in $vals; do
  export $var=$v
  timeout -k 10 200 python bench.py --fastq --fastq-resident-only --steps 20 --warmup 3 > gpurun_out/ab_fq.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_fq.log").read().strip().splitlines()[-1])
print("$var=$v", d["results"]["gpu_tokenizer_text_resident_in_hbm"])
PY
done; done
