# usage: bash tools/bench_all.sh  -- one line per workload: value, ms/step, tile kernel GB/s, tile kernel ms, finalize ms
for w in config2 config4 config4adv config3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel_ms'], d['config']['finalize_ms'])"
done
