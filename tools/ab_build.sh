#!/bin/bash
# A/B builds of the library: tools/ab_build.sh NAME [extra compiler flags]  ->  ab/libptmi_NAME.so  (run with PTMI_LIB=ab/libptmi_NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab
cp -f ptmcmcsampler_amd/libptmi.so /tmp/libptmi_keep.so 2>/dev/null || true
PTMI_EXTRA_CXXFLAGS="$*" python -c "from ptmcmcsampler_amd import _build; _build.build(force=True)"
mv ptmcmcsampler_amd/libptmi.so ab/libptmi_$name.so
cp -f /tmp/libptmi_keep.so ptmcmcsampler_amd/libptmi.so 2>/dev/null || true
touch ptmcmcsampler_amd/libptmi.so
echo ab/libptmi_$name.so
