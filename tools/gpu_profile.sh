#!/bin/bash
# rocprofv3 passes for the bench kernels; run on the GPU box through gpurun from the repo root:
#   gpurun --timeout 1800 -- 'bash tools/gpu_profile.sh r04'
# Writes text summaries under gpurun_out/prof_<tag>/ ; copy them to profiles/<tag>_*.txt (tracked).
# One bench "step" = 100 MH iterations + the PT swap (bench.py).  Counter passes never combine with --stats or trace domains.
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    local prof=(); while [ "$1" != "--" ]; do prof+=("$1"); shift; done; shift
    timeout 400 rocprofv3 --kernel-trace "${prof[@]}" -d $OUT/$name -o $name -- python $ROOT/bench.py --no-cpu-baseline --ess-window 0 --also off "$@" > $OUT/$name.log 2>&1
    echo "$name rc=$?"
    # WIN=N:PATTERN: summarise the timed region only (from the N-th last launch of the dominant kernel on: the warm-up launches, with
    # their other cycle composition, cold clocks and -- config 5 -- long first trees, stay out of the averages)
    if [ -n "${WIN:-}" ]; then python $ROOT/tools/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name.txt --window "$WIN" > /dev/null
    else python $ROOT/tools/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name.txt > /dev/null; fi
    rm -rf $OUT/$name              # the rocpd databases are large; only the text summaries travel back
}
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
LDS="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS"
MF="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
# config 2 (the headline): the driver's command, per-kernel times, HBM traffic counters (separate passes), SQ activity
WIN="20:mh_steps_kernel" run scam_stats --stats -- --steps 20 --warmup 5
run scam_fetch --pmc FETCH_SIZE -- --steps 20 --warmup 5
run scam_write --pmc WRITE_SIZE -- --steps 20 --warmup 5
run scam_sq --pmc $SQ -- --steps 20 --warmup 5
run scam_lds --pmc $LDS -- --steps 20 --warmup 5
# config 3 (dense Gaussian, matrix cores): SCAM cycle and the AM-weighted default mix
WIN="20:mh_dense_scam_kernel" run dense_stats --stats -- --logl dense --steps 20 --warmup 10
run dense_sq --pmc $MF -- --logl dense --steps 20 --warmup 5
WIN="12:mh_pc_kernel" run densemix_stats --stats -- --logl dense --mix default --pick walker --steps 12 --warmup 105
# default SCAM/AM/DE mix with DE active (burn = 10000 iterations = 100 steps): per chain pick and per walker pick
WIN="20:mh_pc_kernel" run mix_stats --stats -- --mix default --steps 20 --warmup 105
run mix_sq --pmc $MF -- --mix default --steps 10 --warmup 105
run mix_sq2 --pmc $SQ -- --mix default --steps 10 --warmup 105
WIN="20:mh_" run mixw_stats --stats -- --mix default --pick walker --steps 20 --warmup 105
# per-walker covariance: the device's tridiagonal QL eigensolver (reduce / chain / apply kernels)
WIN="20:mh_steps_kernel" run pwd_stats --stats -- --cov-mode per_walker_device --steps 20 --warmup 10
# config 5 shape on one GPU (curved likelihood, SCAM / DE / NUTS) and config 4's share of one GPU (1000-d, 64 x 512 chains)
WIN="6:mh_steps_gj_kernel" run c5_stats --stats -- --logl curved --ndim 20 --ntemps 16 --mix nuts --steps 6 --warmup 4
WIN="30:mh_steps_kernel" run c4_stats --stats -- --ndim 1000 --nwalkers 512 --steps 30 --warmup 20
run c4mix_stats --stats -- --ndim 1000 --nwalkers 512 --mix default --steps 3 --warmup 1
ls $OUT/*.txt
# vector / LDS activity of the per-walker epoch kernels (Welford rows, QL reduce / chains / register apply) and of config 4's epoch
# (step kernel, statistics, sytrd_lds_kernel): counters only, one pass each
run pwd_sq --pmc $SQ -- --cov-mode per_walker_device --steps 20 --warmup 10
run pwd_lds --pmc $LDS -- --cov-mode per_walker_device --steps 20 --warmup 10
run c4_sq --pmc $SQ -- --ndim 1000 --nwalkers 512 --steps 30 --warmup 20
ls $OUT/*.txt
