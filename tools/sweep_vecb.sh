# tile kernel GB/s for both boundary strategies over mean read lengths (lognormal sigma 1.1774: mean = 2 e^mu)
for mean in 150 300 600 1200 2400 4800 9600; do
  mu=$(python -c "import math; print(round(math.log($mean/2.0),4))")
  reads=$(python -c "print(min(4000000, int(3.0e9/$mean)))")
  for v in 0 1; do
    PPQC_TILE_VECB=$v timeout -k 10 200 python bench.py --no-cpu-baseline --mu $mu --reads $reads --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mean', $mean, 'vecb', $v, 'GB/s', d['roofline']['achieved'], 'kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'])"
  done
done
