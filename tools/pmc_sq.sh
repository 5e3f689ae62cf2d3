# One rocprofv3 --pmc pass of eight SQ counters over tools/tile_ab.py (GPU box, repo root).
# usage: bash tools/pmc_sq.sh <workload> <reads> <variants> <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
AB_WORKLOAD=$1 AB_VARIANTS=$3 AB_SHIFT=${5:-16} timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/$4 -- python3 tools/tile_ab.py $2 1 > gpurun_out/$4.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/$4/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "tile_kernel" in k or "dense_kernel" in k:
        print(k, {c: round(v / cnt[(k, c)] / 1e6, 2) for c, v in d.items()}, "(millions per launch)")
PY
