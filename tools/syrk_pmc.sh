#!/bin/bash
# counters of the pooled-statistics kernels (developer tool): bash tools/syrk_pmc.sh <tag> [syrk_timing args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
python $ROOT/tools/syrk_timing.py "$@" 2>&1 | tail -2
i=0
for PMC in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/sp$i -o sp -- python $ROOT/tools/syrk_timing.py "$@" > /dev/null 2>&1
  python $ROOT/tools/rocpd_summary.py /tmp/sp$i/sp_results.db $ROOT/gpurun_out/syrk_${TAG}_$i.txt > /dev/null
  grep -E "pool_syrk" $ROOT/gpurun_out/syrk_${TAG}_$i.txt | cut -c1-40,86-150
done
