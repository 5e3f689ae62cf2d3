# A/B of the tile-kernel variants over read-length regimes, one process per workload (GPU box, repo root).
# 0: the shipped pair (tile_kernel + dense_kernel, per-tile probe); 0s / 0d: every tile forced sparse (= the round-1
# default kernel) / dense; 0a: dense_kernel alone over every tile.
# usage: bash tools/ab_uni.sh [variants] [shifts]
vars=${1:-0,0s,0d,0a}
shifts=${2:-16,14}
for w in "mean150 8000000" "mean250 8000000" "mean500 4000000" "mean1000 3000000" "mean2500 1500000" "config2 500000" "adv 200000"; do
  set -- $w
  AB_WORKLOAD=$1 AB_VARIANTS=$vars AB_SHIFT=$shifts timeout -k 10 280 python -u tools/tile_ab.py $2 3 || exit 1
done
