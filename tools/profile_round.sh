# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <outdir under gpurun_out> [bench.py arguments]
# Four separate rocprofv3 runs of bench.py (config 2 unless the extra arguments choose another workload): kernel trace +
# stats, one PMC pass each for FETCH_SIZE and WRITE_SIZE, and one pass of eight SQ counters (counters are never combined
# with tracing).  Summarise afterwards with tools/summarize_prof.py.
out=${1:-gpurun_out/prof}
shift
extra="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-h2d $extra > $out.kt.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-h2d $extra > $out.fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-h2d $extra > $out.write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $out/sq -- python3 bench.py --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-h2d $extra > $out.sq.log 2>&1 || exit 1
tail -1 $out.kt.log | cut -c1-600
