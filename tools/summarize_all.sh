# usage (here, after tools/profile_all.sh <tag> on the GPU box): bash tools/summarize_all.sh <tag>
tag=${1:-r03}
for leg in c2 m500 adv c3 fastq fastq500; do
  echo "== $leg"; python tools/summarize_prof.py gpurun_out/${tag}_$leg profiles/${tag}_bench_$leg | tail -4
done
