# usage: tools/_ab.sh "lib1 lib2 ..." ; runs AM-only / mix timings with each library (PTMI_LIB)
B="python bench.py --no-cpu-baseline"
run() { name=$1; shift; $B "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '%.4g'%j['value'], j['roofline']['avg_launch_ms'])"; }
for lib in $1; do
  export PTMI_LIB=$lib
  [ "$lib" = default ] && unset PTMI_LIB
  echo "== $lib"
  python -m pytest tests/test_gpu_bench_kernels.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -2
  run am_only --weights 0,20,0 --steps 30 --warmup 10
  run mix_chain --mix default --steps 60 --warmup 110
  run mix_walker --mix default --pick walker --steps 60 --warmup 110
done
