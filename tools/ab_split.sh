#!/bin/bash
# A/B builds of the split path's row kernels only: tools/ab_split.sh NAME [-DFLAG ...] -> ptmcmcsampler_amd/csrc/build/ab_NAME.so
# (only ptmi_split.hip is recompiled, the other objects of the last full build are linked as they are; run with PTMI_LIB=<that file>)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
B=ptmcmcsampler_amd/csrc/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c ptmcmcsampler_amd/csrc/ptmi_split.hip -o $B/split_$name.o
objs=$(ls $B/*.o | grep -v '/split' )
hipcc --offload-arch=gfx950 -shared -fPIC -o $B/ab_$name.so $objs $B/split_$name.o
rm -f $B/split_$name.o
echo $B/ab_$name.so
