for v in 3 6 10 12 106 108 110; do
 for mean in 1200 4800 20000; do
  mu=$(python -c "import math; print(round(math.log($mean/2.0),4))")
  reads=$(python -c "print(min(4000000, int(6.0e9/$mean)))")
  PPQC_TILE_VARIANT=$v timeout -k 10 200 python bench.py --no-cpu-baseline --mu $mu --reads $reads --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant', $v, 'mean', $mean, 'GB/s', d['roofline']['achieved'], 'kernel_ms', d['roofline']['kernel_ms'])"
 done
done
