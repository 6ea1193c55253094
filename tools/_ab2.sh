# usage: tools/_ab2.sh "lib1 lib2 ..." ; timing only (no parity) of scam / am / dense with each library
B="python bench.py --no-cpu-baseline"
run() { name=$1; shift; $B "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '%.4g'%j['value'], j['roofline']['avg_launch_ms'])"; }
for lib in $1; do
  export PTMI_LIB=$lib
  [ "$lib" = default ] && unset PTMI_LIB
  echo "== $lib"
  run scam --steps 100 --warmup 20
  run scam --steps 100 --warmup 20
  run am_only --weights 0,20,0 --steps 30 --warmup 10
  run dense --logl dense --steps 40 --warmup 10
done
