# A/B of library builds on a bench leg, alternating runs in one call (one box).
# usage (GPU box, repo root): bash tools/ab_lib.sh "libA.so libB.so ..." reps [bench.py arguments]     ("-" = the in-tree library)
libs=$1; reps=${2:-3}; shift 2
for rep in $(seq $reps); do for l in $libs; do
  if [ "$l" = "-" ]; then unset PPQC_LIB_PATH; else export PPQC_LIB_PATH=$PWD/$l; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d "$@" > gpurun_out/ab_lib.log 2>&1 || { tail -5 gpurun_out/ab_lib.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_lib.log").read().strip().splitlines()[-1])
print("$l", "ms/step", d.get("ms_per_step"), "kernel_ms", d.get("roofline", {}).get("kernel_ms"), "GB/s", d.get("roofline", {}).get("achieved"), d.get("results", ""))
PY
done; done
