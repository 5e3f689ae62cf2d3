# A/B of library builds on the resident FASTQ leg, alternating runs in one call (one box).
# usage (GPU box, repo root): bash tools/ab_fastq.sh "libA.so libB.so ..." [reps]     ("-" = the in-tree library)
libs=$1; reps=${2:-3}
for rep in $(seq $reps); do for l in $libs; do
  if [ "$l" = "-" ]; then unset PPQC_LIB_PATH; else export PPQC_LIB_PATH=$PWD/$l; fi
  timeout -k 10 200 python bench.py --fastq --fastq-resident-only --steps 20 --warmup 3 > gpurun_out/ab_fq.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_fq.log").read().strip().splitlines()[-1])
print("$l", d["results"]["gpu_tokenizer_text_resident_in_hbm"])
PY
done; done
