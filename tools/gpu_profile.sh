#!/bin/bash
# rocprofv3 passes for the bench kernels; run on the GPU box through gpurun from the repo root:
#   gpurun --timeout 1200 -- 'bash tools/gpu_profile.sh r01'
# Writes rocpd databases + text summaries under gpurun_out/prof_<tag>/ ; the summaries are copied to profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    local prof=(); while [ "$1" != "--" ]; do prof+=("$1"); shift; done; shift
    timeout 400 rocprofv3 --kernel-trace "${prof[@]}" -d $OUT/$name -o $name -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/$name.log 2>&1
    echo "$name rc=$?"
    python $ROOT/tools/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name.txt > /dev/null
    rm -rf $OUT/$name              # the rocpd databases are large; only the text summaries travel back
}
# config 2 (the headline): per-kernel times, HBM traffic counters (separate passes), SQ activity
run scam_stats --stats --        # the default bench command: the line the driver records
run scam_fetch --pmc FETCH_SIZE -- --steps 1000 --warmup 200
run scam_write --pmc WRITE_SIZE -- --steps 1000 --warmup 200
run scam_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -- --steps 1000 --warmup 200
# config 3 (dense Gaussian, matrix cores) and the default SCAM/AM/DE mix
run dense_stats --stats -- --logl dense --steps 400 --warmup 100
run dense_sq --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -- --logl dense --steps 300 --warmup 100
run mix_stats --stats -- --mix default --steps 1000 --warmup 100
# config 5 shape on one GPU (curved likelihood, SCAM / DE / NUTS) and the covariance epoch kernels
run c5_stats --stats -- --logl curved --ndim 20 --ntemps 16 --mix nuts --steps 300 --warmup 100
run welford_sq --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -- --steps 3000 --warmup 200
ls $OUT/*.txt
