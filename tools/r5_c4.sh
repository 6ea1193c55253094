#!/bin/bash
# round-5 config-4 step kernel: parity tests of the wide shapes, then A/B timing of the register budgets (developer tool)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide or ragged or full_cycle or pooled_covariance or am_increments" 2>&1 | tail -5
python -m pytest tests/test_am_rle_gpu.py tests/test_gpu_bench_kernels.py -m gpu -q -x -k "rle_engine or checkpoint or side_stream or config4" 2>&1 | tail -5
bash tools/ab_run.sh "--ndim 1000 --nwalkers 512 --steps 30 --warmup 20" w2 w3 w5 2>&1 | tee gpurun_out/r5_c4_ab.txt
