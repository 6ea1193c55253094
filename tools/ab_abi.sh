#!/bin/bash
# A/B build of the ABI unit (ptmi_abi.hip: statistics, eigensolvers, swap kernels): tools/ab_abi.sh NAME [extra flags]
#   -> ptmcmcsampler_amd/libptmi_NAME.so (run with PTMI_LIB=...).  The shape units are those of the last full build: NOT after a change of
#   ptmi_common.h (ptmi_engine / KArgs layouts: the units would disagree about them -- do a full build then).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
B=ptmcmcsampler_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c ptmcmcsampler_amd/csrc/ptmi_abi.hip -o /tmp/ab_abi_$name.o
objs=$(ls $B/*.o | grep -v "/abi.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ptmcmcsampler_amd/libptmi_$name.so $objs /tmp/ab_abi_$name.o
echo ptmcmcsampler_amd/libptmi_$name.so
