#!/bin/bash
# SQ / LDS / MFMA counters of the dense default mix (mh_pc_kernel<25,1,0,true>): gpurun -- 'bash tools/r6_densemix_pmc.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
run() { local name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- python $ROOT/bench.py --logl dense --mix default --pick walker --steps 6 --warmup 105 --no-cpu-baseline --ess-window 0 --also off > $OUT/$name.log 2>&1
  python $ROOT/tools/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name.txt > /dev/null; rm -rf $OUT/$name; grep "mh_pc_kernel" $OUT/$name.txt | grep -v "^void mh_pc_kernel.* [0-9]* *[0-9]* *[0-9.]* *[0-9]* *[0-9]* *[0-9.]*$" | cut -c1-60,95-160; }
run densemix_sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU
run densemix_lds SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
run densemix_mf SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run densemix_mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES
