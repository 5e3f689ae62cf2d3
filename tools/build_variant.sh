# Build a variant of the library from the tree's sources with extra compiler flags (e.g. -DPPQC_X=1) into tools/ab_libs/<name>.so
# usage: bash tools/build_variant.sh <name> [flags]
name=$1; shift
mkdir -p tools/ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function "$@" -o tools/ab_libs/$name.so pypore_amd/csrc/ppqc_api.hip pypore_amd/csrc/ppqc_tile.hip pypore_amd/csrc/ppqc_tables.hip pypore_amd/csrc/ppqc_sort.hip pypore_amd/csrc/ppqc_synth.hip pypore_amd/csrc/ppqc_fastq_gpu.hip pypore_amd/csrc/ppqc_fastq1.hip pypore_amd/csrc/ppqc_fastq.cpp
