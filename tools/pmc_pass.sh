# One rocprofv3 --pmc pass over bench.py (GPU box, repo root); prints per-kernel per-launch means in millions.
# usage: bash tools/pmc_pass.sh <outdir under gpurun_out> "<counters>" [bench.py arguments]
out=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-h2d "$@" > gpurun_out/$out.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ppqc::", "")[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items()):
    print("%-40s" % k, {c: round(v / cnt[(k, c)] / 1e6, 3) for c, v in d.items()})
PY
