#!/bin/bash
# A/B build of ONE shape unit: tools/ab_unit.sh NAME G E L PART [extra flags]  ->  ptmcmcsampler_amd/libptmi_NAME.so (travels with gpurun;
# run with PTMI_LIB=ptmcmcsampler_amd/libptmi_NAME.so).  The other objects are those of the last full build.
set -e
cd "$(dirname "$0")/.."
name=$1; G=$2; E=$3; L=$4; PART=$5; shift 5
B=ptmcmcsampler_amd/csrc/build
if [ $PART = 1 ]; then obj=shape_full_${G}_${E}_${L}.o; sched=""; else obj=shape_${G}_${E}_${L}.o; sched="-mllvm -amdgpu-sched-strategy=max-ilp"; fi
track=""; [ $L != 2 ] && track="-mllvm -amdgpu-use-amdgpu-trackers"
[ $L = 2 ] && sched=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DPTMI_G=$G -DPTMI_E=$E -DPTMI_L=$L -DPTMI_PART=$PART $track $sched "$@" \
    -c ptmcmcsampler_amd/csrc/ptmi_shape.hip -o /tmp/ab_$name.o
objs=$(ls $B/*.o | grep -v "/$obj$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ptmcmcsampler_amd/libptmi_$name.so $objs /tmp/ab_$name.o
echo ptmcmcsampler_amd/libptmi_$name.so
