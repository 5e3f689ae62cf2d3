for cfg in "512 256" "256 256" "1024 256" "2048 256" "256 1024" "512 1024" "128 1024" "256 512"; do
  set -- $cfg
  PPQC_TAB_BLOCKS=$1 PPQC_TAB_THREADS=$2 PPQC_EXT_BLOCKS=$1 PPQC_EXT_THREADS=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/ab_tab.log 2>&1 || exit 1
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tab.log').read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["config"]["finalize_ms"], d["config"]["sort_ms"], d["roofline"]["kernel_ms"])
PY
done
