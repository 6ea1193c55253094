#!/bin/bash
# The bench lines DESIGN.md quotes, one JSON line per workload, into gpurun_out/sweep_<tag>/ (copy the merged file to
# profiles/<tag>_bench.json).  usage: gpurun --timeout 2400 -- 'bash tools/bench_sweep.sh r04'
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sweep_$TAG
mkdir -p $OUT
cd $ROOT
b() { local name=$1; shift; python bench.py --also off "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
python bench.py --steps 20 --warmup 5 > $OUT/driver.json 2> $OUT/driver.err; echo "driver rc=$?"   # exactly what the driver runs (CPU baseline + the legs of the other configs under "also")
b scam --no-cpu-baseline                                         # config 2, default length (200 steps after 100)
b mix_chain --no-cpu-baseline --mix default --steps 100 --warmup 110
b mix_walker --no-cpu-baseline --mix default --pick walker --steps 100 --warmup 110
b dense --no-cpu-baseline --logl dense --steps 50 --warmup 20 --ess-window 0   # config 3, SCAM cycle
b dense_mix_walker --no-cpu-baseline --logl dense --mix default --pick walker --steps 30 --warmup 110 --ess-window 0
b dense_mix_chain --no-cpu-baseline --logl dense --mix default --steps 30 --warmup 110 --ess-window 0
b scam_rows_nolag --no-cpu-baseline --am-mode rows --eig-lag 0 --ess-window 0        # config 2 as round 3 ran it: every row stored, table applied at once
b per_walker_lapack --no-cpu-baseline --cov-mode per_walker --steps 20 --warmup 10 --ess-window 0
b per_walker_jacobi --no-cpu-baseline --cov-mode per_walker_jacobi --steps 30 --warmup 10 --ess-window 0
b per_walker_ql --no-cpu-baseline --cov-mode per_walker_device --steps 30 --warmup 10 --ess-window 0       # tridiagonal QL on the device
b callback_hip --no-cpu-baseline --callback --callback-kind hip --steps 20 --warmup 5 --ess-window 0          # the split path on contiguous rows, the callback a device kernel
b callback_torch --no-cpu-baseline --callback --callback-kind norm --steps 20 --warmup 5 --ess-window 0       # ... a torch expression (one pass)
b callback_torch_naive --no-cpu-baseline --callback --callback-kind naive --steps 20 --warmup 5 --ess-window 0   # ... a torch expression with a temporary
b callback_two_launches --no-cpu-baseline --callback --callback-kind hip --callback-launches two --steps 20 --warmup 5 --ess-window 0
b callback_mix --no-cpu-baseline --callback --callback-kind hip --mix default --steps 20 --warmup 105 --ess-window 0   # AM increments ahead of the proposals
b callback_long --no-cpu-baseline --callback --callback-kind hip --steps 100 --warmup 400 --ess-window 0     # at the stationary acceptance
b callback_small_graph --no-cpu-baseline --callback --callback-kind norm --callback-graph --nwalkers 16 --steps 20 --warmup 5 --ess-window 0
b callback_small --no-cpu-baseline --callback --callback-kind norm --nwalkers 16 --steps 20 --warmup 5 --ess-window 0
PTMI_SPLIT_ROWS=0 b callback_shape_kernels --no-cpu-baseline --callback --callback-kind naive --callback-launches two --steps 10 --warmup 2 --ess-window 0   # round 5's path
b c4_share --no-cpu-baseline --ndim 1000 --nwalkers 512 --steps 40 --warmup 20 --ess-window 0
b c4_mix --no-cpu-baseline --ndim 1000 --nwalkers 512 --mix default --steps 6 --warmup 2 --ess-window 0
b c5_share --no-cpu-baseline --logl curved --ndim 20 --ntemps 16 --mix nuts --steps 6 --warmup 4 --ess-window 0
b oddeven --no-cpu-baseline --swap-mode oddeven --steps 100 --warmup 20 --ess-window 0
b scam_stats_async --no-cpu-baseline --stats-async on --eig-lag 2 --ess-window 0     # statistics on a side stream (two AM rings): measured, not the default
b c4_share_stats_async --no-cpu-baseline --ndim 1000 --nwalkers 512 --steps 40 --warmup 20 --stats-async on --ess-window 0
PTMI_SYTRD_LIB=1 b c4_share_library_dc --no-cpu-baseline --ndim 1000 --nwalkers 512 --steps 40 --warmup 20 --ess-window 0     # PTMI_SYTRD_LIB set below: rocsolver's dstedc / dormtr (round 4)
python - <<PY
import json, glob, os
out = {}
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        out[os.path.basename(f)[:-5]] = json.load(open(f))
    except Exception as e:
        out[os.path.basename(f)[:-5]] = {"error": str(e)}
json.dump(out, open("$OUT/all.json", "w"), indent=1)
for k, d in out.items():
    if "value" in d:
        r = d["roofline"]
        print("%-20s %.4g upd/s  %.3f ms/step  kernel %.3f ms (%.0f%% of wall)  %s frac %.3f  ess/s %s" % (
            k, d["value"], d["ms_per_step"], r["avg_launch_ms"], 100 * r["kernel_time_share_of_wall"], r["bound"], r["frac"], d.get("ess_per_sec")))
    else:
        print(k, d)
PY
