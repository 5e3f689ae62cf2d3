# usage (on the GPU box, from the repo root): bash tools/profile_all.sh <tag>
# Every leg the round's profiles cover, one tools/profile_round.sh each (four rocprofv3 runs: kernel trace, FETCH_SIZE,
# WRITE_SIZE, eight SQ counters), into gpurun_out/<tag>_<leg>/.  Summarise here afterwards: tools/summarize_all.sh <tag>.
tag=${1:-r03}
run() { leg=$1; shift; rm -rf gpurun_out/${tag}_$leg; echo "== $leg"; bash tools/profile_round.sh gpurun_out/${tag}_$leg "$@" || exit 1; }
run c2 &&
run m500 --mu 5.5214 --reads 4000000 &&
run adv --workload config4adv &&
run c3 --workload config3 &&
run fastq --fastq --fastq-resident-only &&
run fastq500 --reads 2000000 --mu 5.5214 --fastq --fastq-reads 2000000 --fastq-resident-only
