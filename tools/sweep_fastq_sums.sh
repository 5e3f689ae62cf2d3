# Where do the span sums of the scan pass pay?  Resident FASTQ leg at several mean read lengths, second pass
# (--tune fastq_sums=0) against span sums (2), alternating in one call.
# usage (GPU box, repo root): bash tools/sweep_fastq_sums.sh ["mean:reads ..."]
for mr in ${1:-500:200000 1000:200000 2000:200000 4000:200000 8000:200000}; do
  mean=${mr%%:*}; reads=${mr##*:}
  mu=$(python -c "import math; print(math.log($mean/2.0))")
  for v in 0 2 0 2; do
    timeout -k 10 300 python bench.py --fastq --fastq-resident-only --mu $mu --reads $reads --fastq-reads $reads --steps 10 --warmup 2 --tune fastq_sums=$v > gpurun_out/sw_fq.log 2>&1 || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/sw_fq.log").read().strip().splitlines()[-1])
print("mean $mean SUMS=$v", d["results"]["gpu_tokenizer_text_resident_in_hbm"], d["config"]["workload"][:60])
PY
  done
done
