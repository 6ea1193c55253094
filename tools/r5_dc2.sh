#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in libptmi libptmi_dcv2; do
OUT=$(pwd)/gpurun_out/dc_tr; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
PTMI_LIB=$GRAFT_REPO_ROOT/ptmcmcsampler_amd/$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o tr -- python $GRAFT_REPO_ROOT/tools/sytrd_check.py 1000 > $OUT/log.txt 2>&1
echo "== $v"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/tr_results.db /tmp/x.txt | grep -E "sytrd_lds|backtransform|prep_kernel|leaf_kernel|secular|gemm_kernel|zhat|vectors_kernel|rank_kernel|copy_deflated|eig_sort|split_kernel" | grep -v Cijk | cut -c1-40,86-150 | head -14
rm -rf $OUT; cd $GRAFT_REPO_ROOT
done
