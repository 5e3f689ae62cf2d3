# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <outdir under gpurun_out>
# Three separate rocprofv3 runs of bench.py (config 2): kernel trace + stats, then one PMC pass each for FETCH_SIZE and
# WRITE_SIZE (never combined with tracing).  Summarise afterwards with tools/summarize_prof.py.
out=${1:-gpurun_out/prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $out.kt.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d > $out.fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d > $out.write.log 2>&1 || exit 1
tail -1 $out.kt.log | cut -c1-300
