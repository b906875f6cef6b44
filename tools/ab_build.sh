#!/bin/bash
# Dev tool: a second build of liblasr.so for same-call A/B runs (LASR_LIB_PATH=build_ab/liblasr_<tag>.so).
# usage: bash tools/ab_build.sh <tag> <file.hip> "<extra hipcc flags, e.g. -DLASR_DW_NO_BFRAG_DPP>"   (the other objects come from build/)
tag=$1; src=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_ab
base=$(basename $src .hip)
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c $root/lightning_asr_amd/csrc/$base.hip -o $root/build_ab/${base}_$tag.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_ab/liblasr_$tag.so $(ls $root/build/*.o | grep -v "/$base.o") $root/build_ab/${base}_$tag.o || exit 1
echo "built build_ab/liblasr_$tag.so"
