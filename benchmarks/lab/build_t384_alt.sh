#!/bin/bash
# Builds benchmarks/lab/alt/libdvd_t384_<tag>.so = the LAB library with gemm_t384_body.inc regenerated under an experiment
# switch of the generator (e.g. T384_PIECES=0,2,4,6,8), for A/B runs through `--lib` (benchmarks/_lab.py).  The tree is copied
# to /tmp so the committed generated files stay untouched.   usage: bash benchmarks/lab/build_t384_alt.sh <tag> VAR=value ...
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
w=/tmp/t384_alt_$tag; rm -rf $w; mkdir -p $w/dvd_amd $w/benchmarks/lab $w/include
cp -r $root/dvd_amd/csrc $w/dvd_amd/; cp -r $root/benchmarks/lab/csrc $w/benchmarks/lab/; cp $root/include/*.h $w/include/
rm -rf $w/dvd_amd/csrc/obj
( cd $w/dvd_amd/csrc && env "$@" python3 gen_gemm_t384.py && env "$@" python3 gen_gemm_t384.py --lab && make -j6 lab >/dev/null 2>&1 )
mkdir -p $root/benchmarks/lab/alt && cp $w/benchmarks/lab/libdvd_hip_lab.so $root/benchmarks/lab/alt/libdvd_t384_$tag.so
echo "built benchmarks/lab/alt/libdvd_t384_$tag.so ($*)"
